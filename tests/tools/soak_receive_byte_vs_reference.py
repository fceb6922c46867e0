#!/usr/bin/env python3
"""Soak of the CPU restatement of receive_byte (oracle/mercury_oracle.c:morc_receive_byte - what the GPU's receive_byte is tested against)
against the reference's OWN cl_telecom_system::receive_byte (oracle/_ref/libmercury_ref_ts.so, oracle/ref_ts_harness.cc) on the randomised
capture windows of tests/test_receive_byte_vs_reference.py: every integer and double of st_receive_stats, the payload and the cross-call
state. Host-only (test infrastructure on both sides).
usage: python tests/tools/soak_receive_byte_vs_reference.py [windows_per_mode=100] [seed=1]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oraclelib import Oracle, RefTelecomSystem  # noqa: E402
from test_receive_byte_vs_reference import ALL_CFGS, compare_one, windows  # noqa: E402


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = total = 0
    t0 = time.time()
    for cfg in ALL_CFGS:
        orc, ref = Oracle(cfg), RefTelecomSystem(cfg)
        rng = np.random.default_rng(seed * 100000 + cfg)
        nbad = dec = trials = 0
        for w, (kind, x, call, state, carrier) in enumerate(windows(orc, rng, W)):
            diff, a, b = compare_one(orc, ref, x, carrier, call, state)
            dec += b["message_decoded"]
            trials += b["sync_trials"]
            if diff:
                nbad += 1
                print("  cfg %d window %d (%s, %s, state %s) differs in %s: oracle %s | reference %s" % (
                    cfg, w, kind, call, state, diff, [a.get(k) for k in diff if k in a], [b.get(k) for k in diff if k in b]))
        print("cfg %3d: %d windows, %d decoded by the reference, %d extra sync trials, %d differ" % (cfg, W, dec, trials, nbad), flush=True)
        bad += nbad
        total += W
        ref.close()
    print("TOTAL %d windows, %d differing (every st_receive_stats field, payload, cross-call state), %.0f s" % (total, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
