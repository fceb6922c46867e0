#!/usr/bin/env python3
"""Phase boundaries of the front-end kernel for one frame from the middle of a full launch (shader-clock stamps, frontend.hip FE_STAMP):
   tools/fe_phases.py [cfg] [frames] [esn0_db]      (needs a GPU; uses torch only for device memory)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C
import torch
from mercury_amd import RxPhy as PhysicalLayer
from mercury_amd.physical_layer import Taps, _ptr

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 8
F = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
es = float(sys.argv[3]) if len(sys.argv) > 3 else 3.5
pl = PhysicalLayer(cfg, max_batch=F)
bb_d = torch.empty(F * pl.frame_samples * 2, dtype=torch.float64, device="cuda")
pl.txgen_dev(1, 0, F, 10 ** (-es / 20), bb_d.data_ptr())
torch.cuda.synchronize()
bb = bb_d.cpu().numpy().view(np.complex128).reshape(F, -1)
names = ["twiddles+FFT", "AGC", "estimate at pilots", "mean |H| (receive_byte only)", "pilot cells", "data cells + variance", "demap", "bit de-interleave + store"]
acc = np.zeros(8)
for rep in range(5):
    cyc = np.zeros(16, np.int64)
    payload = np.zeros((F, pl.payload_stride), np.uint8)
    ts = Taps(**{k: (cyc.ctypes.data if k == "cycles" else None) for k in "grid H eq syms llr_demod llr_ldpc variance agc_gain cycles".split()})
    pl._ck(pl.lib.mgpu_rx_batch_taps(pl.h, _ptr(bb), C.c_int(F), _ptr(payload), None, C.byref(ts)))
    d = np.diff(cyc[:9]).astype(float)
    if rep:
        acc += d
tot = acc.sum()
for n, v in zip(names, acc):
    print("%-32s %8.0f ticks  %5.1f %%" % (n, v / 4, 100 * v / tot))
print("%-32s %8.0f ticks" % ("frame in flight", tot / 4))
