#!/usr/bin/env python3
"""Phase timeline of the LDPC decoder kernels on a real launch (needs a GPU and a library built with -DSPA_STAMPS=1:
   tools/build_variants.sh stamps:"-DSPA_STAMPS=1"; MERCURY_GPU_LIB=mercury_amd/_variants/lib_stamps.so python tools/spa_stamps.py ...).

   tools/spa_stamps.py [decoder spa|spa_fast|minsum] [cfg] [frames] [esn0_db]

Lane 0 of every wavefront of eight workgroups spread over the launch stamps s_memrealtime (100 MHz) at the phase boundaries
(ldpc.hip: SPA_STAMP). Printed: per sampled workgroup its iteration count and the time between boundaries (mean over its
wavefronts, microseconds), and the sum per phase over the sampled workgroups."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mercury_amd import DEC_MINSUM, DEC_SPA, DEC_SPA_FAST, RxPhy

dec = sys.argv[1] if len(sys.argv) > 1 else "spa"
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 8
F = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
es = float(sys.argv[4]) if len(sys.argv) > 4 else 3.5
rx = RxPhy(cfg, max_iters=50, decoder={"spa": DEC_SPA, "spa_fast": DEC_SPA_FAST, "minsum": DEC_MINSUM}[dec], agc=1, variance_source=1, max_batch=F)
if not hasattr(rx.lib, "mgpu_debug_spa_stamps"):
    sys.exit("this library was built without -DSPA_STAMPS=1")
WGS, WAVES, MAXS = 8, 16, 192
bb = torch.empty((F, rx.frame_samples, 2), dtype=torch.float64, device="cuda")
payload = torch.empty((F, rx.payload_stride), dtype=torch.uint8, device="cuda")
stats = torch.empty((F, 6), dtype=torch.int32, device="cuda")
rx.txgen_dev(0x4D455243, 1 << 40, F, float(10.0 ** (-es / 20.0) / np.sqrt(2.0)), bb.data_ptr(), None)
torch.cuda.synchronize()
for _ in range(3):
    rx.receive_dev(bb.data_ptr(), F, payload.data_ptr(), stats.data_ptr())
torch.cuda.synchronize()
buf = np.zeros(WGS * WAVES * MAXS, np.uint64)
rx.lib.mgpu_debug_spa_stamps(None, 1)
rx.enable_timing(True)
rx.receive_dev(bb.data_ptr(), F, payload.data_ptr(), stats.data_ptr())
torch.cuda.synchronize()
fe_ms, dec_ms, _ = rx.kernel_ms_avg()
rx.lib.mgpu_debug_spa_stamps(C.c_void_p(buf.ctypes.data), 0)
buf = buf.reshape(WGS, WAVES, MAXS)
it_all = stats[:, 0].clamp(max=50).cpu().numpy()
names = {(1, 2): "load LLRs, zero messages, fetch records", (2, 3): "syndrome pass", (3, 4): "barrier (syndrome)", (4, 5): "check pass", (8, 5): "check pass",
         (2, 5): "check pass", (5, 6): "barrier (check pass)", (6, 7): "variable update", (7, 8): "barrier (variable update)", (8, 3): "syndrome pass",
         (4, 9): "loop exit", (6, 9): "loop exit", (9, 10): "hard decisions + tail"}
total = {}
out = {"decoder": dec, "cfg": cfg, "frames": F, "esn0_db": es, "decoder_kernel_ms": dec_ms, "frontend_kernel_ms": fe_ms,
       "avg_iterations": float(it_all.mean()), "workgroups": []}
per = F // WGS
for w in range(WGS):
    blk = w * per + F // (2 * WGS)
    waves = [buf[w, v][buf[w, v] != 0] for v in range(WAVES)]
    waves = [x for x in waves if len(x) > 1]
    if not waves:
        continue
    n = min(len(x) for x in waves)
    codes = (waves[0][:n] >> np.uint64(56)).astype(int)
    t = np.stack([(x[:n] & np.uint64((1 << 56) - 1)).astype(np.int64) for x in waves]).astype(float)
    d = np.diff(t, axis=1).mean(axis=0) / 100.0        # us
    span = (t[:, -1].max() - t[:, 0].min()) / 100.0
    phases = {}
    for i in range(n - 1):
        k = names.get((codes[i], codes[i + 1]), "%d->%d" % (codes[i], codes[i + 1]))
        phases[k] = phases.get(k, 0.0) + float(d[i])
        total[k] = total.get(k, 0.0) + float(d[i])
    out["workgroups"].append({"block": int(blk), "iterations": int(it_all[blk]), "wavefronts": len(waves), "span_us": span, "phases_us": phases})
    print("block %5d: %2d iterations, %6.1f us in flight | " % (blk, it_all[blk], span) + ", ".join("%s %.1f" % kv for kv in phases.items()))
tot = sum(total.values())
print("-- sum over the %d sampled workgroups (%.1f iterations on average; launch: %.3f ms for %d frames, %.2f iterations per frame)" %
      (len(out["workgroups"]), np.mean([g["iterations"] for g in out["workgroups"]]) if out["workgroups"] else 0, dec_ms, F, it_all.mean()))
for k, v in sorted(total.items(), key=lambda kv: -kv[1]):
    print("   %-42s %8.1f us  %5.1f %%" % (k, v, 100 * v / tot))
out["phase_share"] = {k: v / tot for k, v in total.items()}
print(json.dumps(out))
