#!/usr/bin/env python3
"""Which of spa_math.h's case branches the wavefronts of a real launch enter (needs a GPU and a library built with -DSPA_CENSUS_ON=1:
   tools/build_variants.sh census:"-DSPA_CENSUS_ON=1"; MERCURY_GPU_LIB=mercury_amd/_variants/lib_census.so python tools/spa_census.py ...).

   tools/spa_census.py [cfg] [frames] [esn0_db] [baseband|receive_byte]

Every branch counts the wavefronts entering it and the lanes they enter it with (ldpc.hip: SPA_CENSUS). Printed per branch: the fraction of
tanh / atanh calls (one call = one wavefront working one 64-slot bin, or 64 variables of the start-up pass) whose instruction stream
contains the branch, and the fraction of lanes that needed it: the gap between the two is what the wavefront pays for its lanes' spread
over fdlibm's cases."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mercury_amd import DEC_SPA, RxPhy

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 8
F = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
es = float(sys.argv[3]) if len(sys.argv) > 3 else -15.0
agc, vs = (1, 1) if (sys.argv[4] if len(sys.argv) > 4 else "receive_byte") == "receive_byte" else (0, 0)      # bench.py's default variant
rx = RxPhy(cfg, max_iters=50, decoder=DEC_SPA, agc=agc, variance_source=vs, max_batch=F)
if not hasattr(rx.lib, "mgpu_debug_spa_census"):
    sys.exit("this library was built without -DSPA_CENSUS_ON=1")
bb = torch.empty((F, rx.frame_samples, 2), dtype=torch.float64, device="cuda")
payload = torch.empty((F, rx.payload_stride), dtype=torch.uint8, device="cuda")
stats = torch.empty((F, 6), dtype=torch.int32, device="cuda")
rx.txgen_dev(0x4D455243, 1 << 40, F, float(10.0 ** (-es / 20.0) / np.sqrt(2.0)), bb.data_ptr(), None)
torch.cuda.synchronize()
rx.lib.mgpu_debug_spa_census(None, C.c_int(1))
rx.receive_dev(bb.data_ptr(), F, payload.data_ptr(), stats.data_ptr())
torch.cuda.synchronize()
buf = np.zeros(64, np.uint64)
rx.lib.mgpu_debug_spa_census(C.c_void_p(buf.ctypes.data), C.c_int(0))
w, l = buf[:32].astype(float), buf[32:].astype(float)
names = {0: "tanh: calls", 1: "  k == 0 ending", 2: "  k != 0: c, e", 4: "    |x| >= 1, k < 20", 5: "    |x| >= 1, k 20..56", 7: "    |x| < 1, k == 1", 8: "    |x| < 1, k >= 2",
         9: "  1 - 2/(t+2)  (|x| >= 1)", 10: "  -t/(t+2)  (|x| < 1)", 11: "  |x| >= 22 / NaN", 12: "atanh: calls", 13: "  clamp of +-1", 14: "  |x| < 0.5", 15: "  |x| >= 0.5",
         16: "  normalised (y >= 0.41421): u, c, f", 17: "    u >= 2", 18: "    u < 2", 19: "  direct tail", 20: "  normalised tail", 21: "    |f| < 2^-20", 22: "  |x| < 2^-28 (some lanes)", 23: "  every lane +-1 or < 2^-28: evaluation skipped"}
print("cfg %d, %d frames at %.1f dB, mean iterations %.2f" % (cfg, F, es, stats[:, 0].float().mean().item()))
for i, n in names.items():
    base = 0 if i < 12 else 12
    print("%-40s wavefronts %12d (%.4f of calls)   lanes %14d (%.4f of lanes)" % (n, w[i], w[i] / max(w[base], 1), l[i], l[i] / max(l[base], 1)))
