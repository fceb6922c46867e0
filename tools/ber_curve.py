#!/usr/bin/env python3
"""The reference's BER_PLOT_baseband self-simulation (telecom_system.cc:2393-2480: 25 Es/N0 points, 100 frames each on the CPU) on the GPU:
prints EsN0;BER;FER lines like the reference does, for far more frames per point.   usage: ber_curve.py [cfg] [frames_per_point] [decoder]
                                                                        ber_curve.py --passband [cfg] [frames_per_point]
--passband: BER_PLOT_passband_process_main (:2432-2470: the audio path, transmit_byte -> AWGN with delay -> receive_byte; 25 points from
-10 dB in 0.5 dB steps x 100 frames for OFDM, 31 points from -25 dB x 3 frames for MFSK, output power 1 W)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mercury_amd import RxPhy, physical_layer as pl  # noqa: E402


def passband(argv):
    cfg = int(argv[0]) if argv else 8
    n = int(argv[1]) if len(argv) > 1 else 4096
    rx = RxPhy(cfg, max_batch=min(n, 1024))
    pts = np.arange(31) * 1.0 - 25.0 if cfg >= 100 else np.arange(25) * 0.5 - 10.0
    rx.passband_test_esn0(pts[-1:], min(n, 1024), 1500.0, output_power_watt=1.0)
    t0 = time.perf_counter()
    res = rx.passband_test_esn0(pts, n, 1500.0, seed=2024, output_power_watt=1.0)
    dt = time.perf_counter() - t0
    for r in res:
        print("%.1f;%.3e;%.3e;%d" % (r["esn0_db"], r["BER"], r["FER"], r["crc_ok_frames"]))
    print(json.dumps({"cfg": cfg, "mode": "passband", "points": len(res), "frames_per_point": n, "seconds": dt, "frames_per_s": len(res) * n / dt}), file=sys.stderr)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--passband":
        return passband(sys.argv[2:])
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
