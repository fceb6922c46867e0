#!/usr/bin/env python3
"""Decoder kernel time against the SCALE of its input LLRs (needs a GPU): 2048 codewords of Gaussian LLRs of standard deviation s, all 50
iterations, for one mode's code.   tools/ldpc_scale_probe.py [cfg]
A launch whose messages run into the denormal range shows what denormal operands cost the fp64 pipeline."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mercury_amd import DEC_SPA, RxPhy

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
F = 2048
rx = RxPhy(cfg, max_iters=50, decoder=DEC_SPA, max_batch=F)
g = torch.Generator(device="cuda").manual_seed(7)
base = torch.randn((F, 1600), generator=g, device="cuda", dtype=torch.float32)
bits = torch.empty((F, rx.K), dtype=torch.uint8, device="cuda")
iters = torch.empty(F, dtype=torch.int32, device="cuda")
for s in (4.0, 1.0, 0.3, 0.1, 0.03, 0.01, 1e-3, 1e-4, 1e-6, 1e-8, 1e-12, 1e-20, 1e-30):
    llr = (base * s).contiguous()
    rx.ldpc_decode_dev(llr.data_ptr(), F, bits.data_ptr(), iters.data_ptr())
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(4):
        rx.ldpc_decode_dev(llr.data_ptr(), F, bits.data_ptr(), iters.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    ev1.record()
    torch.cuda.synchronize()
    print("cfg %d  sigma %8.1e  %.3f ms per launch  mean iterations %.1f" % (cfg, s, ev0.elapsed_time(ev1) / 4, iters.float().mean().item()))
