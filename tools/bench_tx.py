#!/usr/bin/env python3
"""transmit_byte on the GPU (include/mercury_tx.h): messages/s and kernel times for a batch of F messages, device buffers in
and out, next to the CPU restatement on one core. usage: tools/bench_tx.py [cfg] [F]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from mercury_amd import RxPhy  # noqa: E402
from mercury_amd.physical_layer import BATCH_MESSAGE, NO_FILTER_MESSAGE, SINGLE_MESSAGE  # noqa: E402


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    carrier = 48000.0 * 50.0 / 256 / 4 / 2 + 300
    rx = RxPhy(cfg, max_batch=F)
    total = rx.transmit_frame_samples()
    dev = torch.device("cuda", 0)
    pl = torch.randint(0, 256, (F, rx.payload_bytes), dtype=torch.uint8, device=dev)
    out = torch.empty((F, total), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()                                  # the library works on its own stream
    res = {"cfg": cfg, "messages": F, "samples_per_message": total}
    for name, loc in (("single_message_filtered", SINGLE_MESSAGE), ("no_filter_message", NO_FILTER_MESSAGE), ("arq_batch_filtered", BATCH_MESSAGE)):
        if loc == BATCH_MESSAGE and (F + 2) * total >= 2 ** 31:
            continue
        for _ in range(2):
            rx.transmit_byte_dev(pl.data_ptr(), rx.payload_bytes, F, out.data_ptr(), carrier, message_location=loc)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            rx.transmit_byte_dev(pl.data_ptr(), rx.payload_bytes, F, out.data_ptr(), carrier, message_location=loc)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        res[name] = {"ms_per_batch": dt * 1e3, "messages_per_s": F / dt, "output_GB_per_s": F * total * 8 / dt / 1e9}
    try:
        import oraclelib
        orc = oraclelib.Oracle(cfg)
        msg = pl[0].cpu().numpy().astype(np.int32)
        t0 = time.perf_counter()
        k = 20
        for _ in range(k):
            want = orc.transmit_byte(msg)
        dt = (time.perf_counter() - t0) / k
        res["cpu_port_1core_messages_per_s"] = 1.0 / dt
        rx.transmit_byte_dev(pl.data_ptr(), rx.payload_bytes, F, out.data_ptr(), carrier, message_location=SINGLE_MESSAGE)
        res["gpu_equals_cpu_on_message_0"] = bool(np.array_equal(out[0].cpu().numpy(), want))
    except Exception as e:  # the oracle is optional here
        res["cpu_port"] = "unavailable: %s" % e
    print(json.dumps(res))


if __name__ == "__main__":
    main()
