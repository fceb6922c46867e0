#!/usr/bin/env python3
"""Fine Schmidl-Cox search (step 1 over (preamble + 4) symbols, telecom_system.cc:1014-1018) kernel time per launch: dense kernel (variant 0)
vs the shared-products kernel with 4 / 8 candidates per lane (variants 1 / 2), by windows per launch."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mercury_amd import RxPhy  # noqa: E402

rx = RxPhy(8, max_batch=1)
n = rx.Nofdm * 4 * (rx.preamble_nsymb + 4)
rng = np.random.default_rng(0)
for W in ([int(sys.argv[1])] if len(sys.argv) > 1 else (1, 16, 64, 256, 1024)):
    z = rng.standard_normal((W, n)) + 1j * rng.standard_normal((W, n))
    out = {}
    for v in (0, 1, 2):
        rx.debug_tsync_metric(z, 1, v)
        rx.debug_tsync_metric(z, 1, v)
        out[v] = rx.last_sync_kernel_ms()
    print(json.dumps({"windows": W, "candidates": n - rx.Nofdm * 4 * rx.preamble_nsymb, "dense_ms": out[0], "shared_r4_ms": out[1], "shared_r8_ms": out[2]}), flush=True)
