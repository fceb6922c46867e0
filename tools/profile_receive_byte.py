#!/usr/bin/env python3
"""Device-resident receive_byte on W low-noise mode-`cfg` windows, a few calls: run under `rocprofv3 --kernel-trace --stats` for the
per-kernel breakdown of the chain (tools/collect_profiles.sh sweep does).   usage: profile_receive_byte.py [cfg] [W] [calls]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mercury_amd import RxPhy  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rx = RxPhy(cfg, max_batch=W)
# windows from the library's own passband self-simulation (clean point): frames at the reference's test delay behind noise
_, wins, _ = rx.passband_test_esn0([30.0], W, 1500.0, seed=3, want_windows=True)
import torch  # noqa: E402
d = torch.from_numpy(wins).to("cuda:0")
torch.cuda.synchronize()
rx.receive_byte_dev(d.data_ptr(), W, 1500.0)
ts = []
for _ in range(calls):
    t0 = time.perf_counter()
    out = rx.receive_byte_dev(d.data_ptr(), W, 1500.0)
    ts.append(time.perf_counter() - t0)
print({"cfg": cfg, "windows": W, "ms_per_call": [round(t * 1e3, 3) for t in ts], "decoded": int(out["stats"]["message_decoded"].sum())})
