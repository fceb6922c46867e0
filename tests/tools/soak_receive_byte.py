#!/usr/bin/env python3
"""Soak: the batched GPU receive_byte against the CPU restatement on randomised capture windows (random delay, noise level,
carrier offset, frame / noise-only / two frames), every mode. Prints one line per mode and any window whose integer fields differ.
With a third argument "ref" the checker is the reference's OWN cl_telecom_system::receive_byte (oracle/_ref/libmercury_ref_ts.so,
oracle/ref_ts_harness.cc) instead of the restatement, and the doubles (SNR, frequency offset, metric) must be bit-identical as well.
usage: python tests/tools/soak_receive_byte.py [windows_per_mode] [seed] [ref]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
import oraclelib  # noqa: E402
from mercury_amd import RxPhy  # noqa: E402

INT_FIELDS = ("iterations_done", "crc", "all_zeros", "message_decoded", "delay", "sync_trials", "frame_overflow_symbols")


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    use_ref = len(sys.argv) > 3 and sys.argv[3] == "ref"
    bad = 0
    for cfg in list(range(17)) + [100, 101, 102]:
        orc = oraclelib.Oracle(cfg)
        checker = oraclelib.RefTelecomSystem(cfg) if use_ref else orc
        rng = np.random.default_rng(seed * 1000 + cfg)
        n = orc.buffer_samples()
        wins, dfs = [], []
        for w in range(W):
            noise = float(10 ** rng.uniform(-3, -0.3))
            x = rng.standard_normal(n) * noise
            kind = rng.choice(["frame", "frame", "frame", "noise", "two", "edge"])
            pl = rng.integers(0, 256, orc.payload_bytes)
            pb = orc.transmit_byte(pl.astype(np.int32), message_location=int(rng.choice([3, 4]))) * float(rng.uniform(0.8, 4.0))
            used = (orc.preamble_nsymb + orc.active_nsymb) * orc.Nofdm * 4
            if kind in ("frame", "two"):
                d = int(rng.integers(0, n - used))
                x[d: d + used] += pb[:used]
            if kind == "two" and n > 2 * used + 5000:
                d = int(rng.integers(0, n - used))
                x[d: d + used] += pb[:used]
            if kind == "edge":                               # frame cut off by the end of the window
                d = n - int(rng.integers(used // 4, used))
                x[d:] += pb[: n - d]
            wins.append(x)
        wins = np.stack(wins)
        states = None
        if use_ref:      # the harsher generator of tests/test_receive_byte_vs_reference.py (silence, interferers, clipped and far-off-frequency frames) with link states carried in
            from test_receive_byte_vs_reference import windows as ref_windows
            from mercury_amd.physical_layer import LINK_STATE_DTYPE
            gen = list(ref_windows(orc, rng, W))
            wins = np.stack([g[1] for g in gen])
            states = np.zeros(W, LINK_STATE_DTYPE)
            for w, g in enumerate(gen):
                states[w] = g[3]
        df = float(rng.choice([0.0, 0.0, 2.5, -7.0]))
        rx = RxPhy(cfg, max_batch=W)
        out = rx.receive_byte(wins, oraclelib.CARRIER + df, state=None if states is None else states.copy())
        nbad = 0
        for w in range(W):
            ref = checker.receive_byte(wins[w], carrier=oraclelib.CARRIER + df, state=None if states is None else oraclelib.LinkState(*states[w].tolist()))
            st = out["stats"][w]
            diff = [k for k in INT_FIELDS if st[k] != ref[k]]
            if st["coarse_metric"] != ref["coarse_metric"] or st["freq_offset"] != ref["freq_offset"]:
                diff.append("float")
            if use_ref and (st["snr_db"] != ref["snr_db"] or st["signal_strength_dbm"] != ref["signal_strength_dbm"]):
                diff.append("snr/level")
            if not np.array_equal(out["payload"][w][: orc.payload_bytes], ref["payload"]):
                diff.append("payload")
            if diff:
                nbad += 1
                print("  cfg %d window %d differs in %s: gpu %s  cpu %s" % (cfg, w, diff, [st[k] for k in INT_FIELDS], [ref[k] for k in INT_FIELDS]))
        bad += nbad
        print("cfg %3d: %d windows, %d decoded, %d differ%s" % (cfg, W, int(out["stats"]["message_decoded"].sum()), nbad,
                                                                  " (checker: the reference's cl_telecom_system)" if use_ref else ""), flush=True)
        rx.close()
        if use_ref:
            checker.close()
    print("TOTAL differing windows:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
