#!/usr/bin/env python3
"""Bit-exactness soak of the GPU receive path against the CPU oracle at scale (test infrastructure: the oracle is the checker).
For every mode, at its operating point and 1.5 dB below it (inside the waterfall, where iteration counts spread over 1..50 and
some frames fail), F frames generated on the device go through mgpu_rx_batch_dev with the reference decoder (fp64 sum-product)
and through the oracle on all host cores; payload bytes, iteration count and CRC of every frame must agree.
  python tests/tools/parity_soak.py [frames_per_point=2048] [cfg,cfg,...]  > gpurun_out/parity_soak.json      (GPU box)"""
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oraclelib  # noqa: E402
from conftest import OPERATING_ESN0  # noqa: E402
from mercury_amd import DEC_SPA, RxPhy  # noqa: E402

SEED = 0x50415249


def oracle_many(cfg, bb, flags, cores):
    n = bb.shape[0]
    per = (n + cores - 1) // cores
    chunks = [(a, min(n, a + per)) for a in range(0, n, per)]
    ctxs = [oraclelib.Oracle(cfg, 50) for _ in chunks]
    res = [None] * len(chunks)

    def work(i):
        a, b = chunks[i]
        res[i] = ctxs[i].rx_many(bb[a:b], flags)

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(chunks))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    iters = np.concatenate([r[1] for r in res])
    crc = np.concatenate([r[2] for r in res])
    pl = np.concatenate([r[3] for r in res])
    return iters, crc, pl


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    cfgs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(17)) + [100, 101, 102]
    cores = len(os.sched_getaffinity(0))
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    out = {"frames_per_point": F, "host_threads": cores, "modes": {}}
    total = bad = 0
    t00 = time.perf_counter()
    for cfg in cfgs:
        zf = cfg in (15, 16)
        agc, vs = (0, 0) if zf else (1, 1)
        flags = oraclelib.FLAGS_BASEBAND_TEST if zf else oraclelib.FLAGS_RECEIVE_BYTE
        rx = RxPhy(cfg, max_iters=50, decoder=DEC_SPA, agc=agc, variance_source=vs, device=0, max_batch=F)
        bb = torch.empty((F, rx.frame_samples, 2), dtype=torch.float64, device=dev)
        m = {}
        for label, off in (("operating", 0.0), ("waterfall", -1.5)):
            esn0 = OPERATING_ESN0[cfg] + off
            amp = float(10.0 ** (-esn0 / 20.0) / np.sqrt(2.0))
            rx.txgen_dev(SEED, (cfg + 1) << 32, F, amp, bb.data_ptr(), None, stream=stream)
            payload = torch.zeros((F, rx.payload_stride), dtype=torch.uint8, device=dev)
            stats = torch.zeros((F, 6), dtype=torch.int32, device=dev)
            rx.receive_dev(bb.data_ptr(), F, payload.data_ptr(), stats.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            g_pl, g_st = payload.cpu().numpy(), stats.cpu().numpy()
            bb_h = bb.cpu().numpy().view(np.complex128).reshape(F, -1)
            t0 = time.perf_counter()
            iters, crc, pl = oracle_many(cfg, bb_h, flags, cores)
            dt = time.perf_counter() - t0
            mism = (g_st[:, 0] != iters) | (g_st[:, 1] != crc) | (g_pl[:, : pl.shape[1]] != pl).any(axis=1)
            m[label] = {"esn0_db": esn0, "mismatching_frames": int(mism.sum()), "decoded_fraction": float((g_st[:, 3] != 0).mean()),
                        "iterations_min_mean_max": [int(np.minimum(iters, 50).min()), float(np.minimum(iters, 50).mean()), int(np.minimum(iters, 50).max())],
                        "frames_not_converged": int((iters > 50).sum()), "cpu_s": round(dt, 2)}
            total += F
            bad += int(mism.sum())
            print("cfg %3d %-9s Es/N0 %6.1f: %d frames, %d mismatching; decoded %.4f, iterations %d..%d (mean %.1f), %d not converged; CPU %.1f s" % (
                cfg, label, esn0, F, int(mism.sum()), m[label]["decoded_fraction"], *m[label]["iterations_min_mean_max"][::2],
                m[label]["iterations_min_mean_max"][1], m[label]["frames_not_converged"], dt), file=sys.stderr)
        out["modes"][str(cfg)] = m
        rx.close()
    out["total_frames"] = total
    out["total_mismatching_frames"] = bad
    out["wall_s"] = round(time.perf_counter() - t00, 1)
    print("TOTAL %d frames, %d mismatching (payload bytes, iteration count, CRC), %.0f s" % (total, bad, out["wall_s"]), file=sys.stderr)
    print(json.dumps(out, indent=1))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
