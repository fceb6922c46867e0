#!/usr/bin/env python3
"""Does running the front-end of batch n+1 beside the decoder of batch n (two streams, LLR double buffer) buy throughput?
   tools/pipeline_probe.py [decoder] [esn0] [cfg] [frames] [steps]        (GPU box)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mercury_amd import DEC_MINSUM, DEC_SPA, DEC_SPA_FAST, RxPhy

dec = sys.argv[1] if len(sys.argv) > 1 else "spa"
es = float(sys.argv[2]) if len(sys.argv) > 2 else 3.5
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 8
F = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 40
rx = RxPhy(cfg, max_iters=50, decoder={"spa": DEC_SPA, "spa_fast": DEC_SPA_FAST, "minsum": DEC_MINSUM}[dec], max_batch=F)
dev = torch.device("cuda")
bbs = []
for b in range(2):
    bb = torch.empty((F, rx.frame_samples, 2), dtype=torch.float64, device=dev)
    rx.txgen_dev(0x4D455243, b * F, F, float(10.0 ** (-es / 20.0) / np.sqrt(2.0)), bb.data_ptr(), None)
    bbs.append(bb)
payload = [torch.empty((F, rx.payload_stride), dtype=torch.uint8, device=dev) for _ in range(2)]
stats = [torch.empty((F, 6), dtype=torch.int32, device=dev) for _ in range(2)]
llr = [torch.empty((F, 1600), dtype=torch.float32, device=dev) for _ in range(2)]
var = [torch.empty((F,), dtype=torch.float32, device=dev) for _ in range(2)]
torch.cuda.synchronize()


def sequential(n):
    s = torch.cuda.current_stream().cuda_stream
    for i in range(n):
        rx.receive_dev(bbs[i & 1].data_ptr(), F, payload[i & 1].data_ptr(), stats[i & 1].data_ptr(), stream=s)


s_fe, s_dec = torch.cuda.Stream(), torch.cuda.Stream()
ev_fe = [torch.cuda.Event() for _ in range(2)]
ev_dec = [torch.cuda.Event() for _ in range(2)]


def pipelined(n):
    # FE(i) on s_fe into llr[i&1] (after decoder(i-2) has read it); decoder(i) on s_dec after FE(i)
    for i in range(n):
        k = i & 1
        if i >= 2:
            s_fe.wait_event(ev_dec[k])
        rx.frontend_dev(bbs[k].data_ptr(), F, llr[k].data_ptr(), var[k].data_ptr(), stream=s_fe.cuda_stream)
        ev_fe[k].record(s_fe)
        s_dec.wait_event(ev_fe[k])
        rx.ldpc_decode_dev(llr[k].data_ptr(), F, d_payload=payload[k].data_ptr(), d_stats=stats[k].data_ptr(), d_variance=var[k].data_ptr(), stream=s_dec.cuda_stream)
        ev_dec[k].record(s_dec)


for name, fn in (("sequential", sequential), ("pipelined", pipelined), ("sequential", sequential), ("pipelined", pipelined)):
    fn(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    it = float(stats[0][:, 0].clamp(max=50).sum().item()) / F
    print("%-10s %s cfg %d Es/N0 %+5.1f: %.4f ms per step, %.0f frames/s (%.2f iterations per frame)" % (name, dec, cfg, es, dt / steps * 1e3, F * steps / dt, it))
