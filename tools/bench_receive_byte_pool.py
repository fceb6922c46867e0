#!/usr/bin/env python3
"""receive_byte on W capture windows through one context and through pools of 2 / 3 contexts time-sharing ONE GPU (mgpu_pool with
devices = {0, 0[, 0]}): the contexts' host threads, streams and workspaces are independent, so one shard's control rounds and decoder
stragglers run beside the other's kernels. Windows come from the transmit chain (mgpu_transmit_byte_batch) at random delays.
  python tools/bench_receive_byte_pool.py [cfg=8] [W=1024]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mercury_amd import RxPhy, RxPool  # noqa: E402

CARRIER = 48000.0 * 50.0 / 256 / 4 / 2 + 300


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    rx = RxPhy(cfg, max_batch=W)
    n = rx.receive_buffer_samples()
    rng = np.random.default_rng(1)
    msgs = rng.integers(0, 256, (W, rx.payload_bytes), dtype=np.uint8)
    audio = rx.transmit_byte(msgs, CARRIER)
    wins = rng.standard_normal((W, n)) * 0.01
    for w in range(W):
        d = int(rng.integers(5 * 1088, n - audio.shape[1] - 5 * 1088))
        wins[w, d: d + audio.shape[1]] += audio[w]
    dwin = torch.from_numpy(wins).to("cuda:0")
    torch.cuda.synchronize()

    def rate(fn, reps=4):
        fn()
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = fn()
            ts.append(time.perf_counter() - t0)
        return W / sorted(ts)[len(ts) // 2], out

    res = {"cfg": cfg, "windows": W}
    res["one_context_host"], ref = rate(lambda: rx.receive_byte(wins, CARRIER))
    res["one_context_device"], ref_d = rate(lambda: rx.receive_byte_dev(dwin.data_ptr(), W, CARRIER))
    res["decoded"] = int(ref["stats"]["message_decoded"].sum())
    for k in (2, 3, 4):
        pool = RxPool(cfg, [0] * k, max_batch=(W + k - 1) // k)
        r_h, out_h = rate(lambda: pool.receive_byte(wins, CARRIER))
        r_d, out_d = rate(lambda: pool.receive_byte(dwin.data_ptr(), CARRIER, W=W))
        same = bool(np.array_equal(out_h["payload"], ref["payload"]) and out_h["stats"].tobytes() == ref["stats"].tobytes() and
                    np.array_equal(out_d["payload"], ref["payload"]) and out_d["stats"].tobytes() == ref["stats"].tobytes())
        res["pool_%d_contexts" % k] = {"host": r_h, "device": r_d, "identical_to_one_context": same}
        pool.close()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
