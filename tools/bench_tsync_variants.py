#!/usr/bin/env python3
"""Coarse Schmidl-Cox metric (step 100) kernel time per launch: staged kernel (variant 0) vs streaming kernel (variant 1), by windows per launch."""
import json
import os
import sys, json, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mercury_amd import RxPhy
rx = RxPhy(8, max_batch=1)
n = rx.Nofdm*85*4
rng = np.random.default_rng(0)
for W in ([int(sys.argv[1])] if len(sys.argv) > 1 else (16, 64, 256, 1024)):
    z = rng.standard_normal((W, n)) + 1j*rng.standard_normal((W, n))
    out = {}
    for v in (0, 1):
        rx.debug_tsync_metric(z, 100, v)
        rx.debug_tsync_metric(z, 100, v)
        out[v] = rx.last_sync_kernel_ms()
    print(json.dumps({'windows': W, 'staged_ms': out[0], 'stream_ms': out[1]}), flush=True)
