#!/usr/bin/env python3
"""Soak: the GPU RX hot path against the CPU restatement on many frames per mode around the decoding threshold, where iteration
counts vary the most. Compares iteration count, CRC and payload bytes exactly. usage: soak_rx_path.py [frames_per_mode] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
import oraclelib  # noqa: E402
from conftest import OPERATING_ESN0  # noqa: E402
from mercury_amd import RxPhy  # noqa: E402


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    bad = 0
    for cfg in list(range(17)) + [100, 101, 102]:
        orc = oraclelib.Oracle(cfg)
        zf = cfg in (15, 16)
        flags = oraclelib.FLAGS_BASEBAND_TEST if zf else oraclelib.FLAGS_RECEIVE_BYTE
        rx = RxPhy(cfg, max_batch=F, agc=0 if zf else 1, variance_source=0 if zf else 1)
        rng = np.random.default_rng(seed + cfg)
        frames = []
        for f in range(F):
            esn0 = OPERATING_ESN0[cfg] - 2.5 + 3.0 * rng.random()          # straddles the waterfall
            frames.append(orc.gen_frame(seed, 10 ** 6 * cfg + f, oraclelib.noise_amp_for(esn0), channel=int(rng.integers(0, 2)))[0])
        bb = np.stack(frames)
        out = rx.receive(bb)
        res = orc.rx_many(bb, flags)
        it_c, crc_c, pl_c = res[1], res[2], res[3]
        st = out["stats"]
        d_it = int((st["iterations_done"] != it_c).sum())
        d_crc = int((st["crc"] != crc_c).sum())
        d_pl = int((out["payload"][:, : pl_c.shape[1]] != pl_c).any(axis=1).sum())
        bad += d_it + d_crc + d_pl
        print("cfg %3d: %d frames, %d decoded, iteration histogram max %d; differ: iterations %d crc %d payload %d"
              % (cfg, F, int(st["message_decoded"].sum()), int(st["iterations_done"].max()), d_it, d_crc, d_pl), flush=True)
        rx.close()
    print("TOTAL differences:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
