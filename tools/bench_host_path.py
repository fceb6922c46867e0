#!/usr/bin/env python3
"""PCIe-inclusive rate of the blocking host-buffer entry point mgpu_rx_batch (pageable and page-locked input), by chunk size.
  python tools/bench_host_path.py [cfg=8] [F=4096] [esn0=-15]      (GPU box)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mercury_amd import RxPhy  # noqa: E402
from mercury_amd.physical_layer import pinned_empty  # noqa: E402


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    esn0 = float(sys.argv[3]) if len(sys.argv) > 3 else -15.0
    rx = RxPhy(cfg, max_batch=F)
    dev = torch.device("cuda:0")
    bb = torch.empty((F, rx.frame_samples, 2), dtype=torch.float64, device=dev)
    rx.txgen_dev(0x4D455243, 0, F, float(10.0 ** (-esn0 / 20.0) / np.sqrt(2.0)), bb.data_ptr(), None, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    host = bb.cpu().numpy().view(np.complex128).reshape(F, -1)
    pin = pinned_empty(host.shape, np.complex128)
    pin[...] = host
    res = {"cfg": cfg, "frames": F, "esn0_db": esn0, "bytes_per_frame": rx.frame_samples * 16}
    # raw copy rates
    for name, src in (("pageable", host), ("pinned", pin)):
        t = torch.from_numpy(src.view(np.float64))
        d = torch.empty_like(bb.view(-1)[: t.numel()]).view(t.shape)
        d.copy_(t); torch.cuda.synchronize()
        t0 = time.perf_counter(); d.copy_(t); torch.cuda.synchronize()
        res["h2d_GBps_" + name] = t.numel() * 8 / (time.perf_counter() - t0) / 1e9

    def timed(src, reps=6):
        rx.receive(src)
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter(); rx.receive(src); best = min(best, time.perf_counter() - t0)
        return F / best

    os.environ["MERCURY_NO_PIPELINE"] = "1"
    res["one_launch_pageable"] = timed(host); res["one_launch_pinned"] = timed(pin)
    del os.environ["MERCURY_NO_PIPELINE"]
    for chunk in (256, 512, 1024, 2048):
        os.environ["MERCURY_RX_CHUNK"] = str(chunk)
        res["chunk%d_pageable" % chunk] = timed(host); res["chunk%d_pinned" % chunk] = timed(pin)
    del os.environ["MERCURY_RX_CHUNK"]
    res["default_pageable"] = timed(host); res["default_pinned"] = timed(pin)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
