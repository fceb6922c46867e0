#!/usr/bin/env python3
"""The reference's BER_PLOT_baseband self-simulation (telecom_system.cc:2393-2480: 25 Es/N0 points, 100 frames each on the CPU) on the GPU:
prints EsN0;BER;FER lines like the reference does, for far more frames per point.   usage: ber_curve.py [cfg] [frames_per_point] [decoder]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mercury_amd import RxPhy, physical_layer as pl  # noqa: E402


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    dec = {"spa": pl.DEC_SPA, "spa_fast": pl.DEC_SPA_FAST, "minsum": pl.DEC_MINSUM}[sys.argv[3] if len(sys.argv) > 3 else "spa"]
    rx = RxPhy(cfg, max_batch=min(n, 65536), agc=0, variance_source=0, decoder=dec)          # the variant baseband_test_EsN0 runs
    pts = np.arange(-12.0, 13.0, 1.0)[:25] + (0.0 if cfg >= 7 else -6.0)
    rx.baseband_test_esn0(pts[:1], min(n, 4096))
    t0 = time.perf_counter()
    res = rx.baseband_test_esn0(pts, n, seed=2024)
    dt = time.perf_counter() - t0
    for r in res:
        print("%.1f;%.3e;%.3e;%.2f" % (r["esn0_db"], r["BER"], r["FER"], r["avg_iterations"]))
    print(json.dumps({"cfg": cfg, "points": len(res), "frames_per_point": n, "seconds": dt, "frames_per_s": len(res) * n / dt}), file=sys.stderr)


if __name__ == "__main__":
    main()
