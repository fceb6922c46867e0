#!/usr/bin/env python3
"""Per-mode soft-output deviation of the HIP front-end from the CPU oracle (record max abs/rel
error per mode). Run on the GPU box: python tests/tools/llr_error_table.py > gpurun_out/llr_error.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oraclelib  # noqa: E402
from conftest import OPERATING_ESN0, SEED  # noqa: E402
from mercury_amd import RxPhy  # noqa: E402


def main():
    out = {}
    for cfg in range(17):
        orc = oraclelib.Oracle(cfg, 50)
        agc, vs, flags = (0, 0, oraclelib.FLAGS_BASEBAND_TEST) if cfg in (15, 16) else (1, 1, oraclelib.FLAGS_RECEIVE_BYTE)
        F = 24
        bb = np.stack([orc.gen_frame(SEED, 5000 + i, oraclelib.noise_amp_for(OPERATING_ESN0[cfg] + (i % 4)))[0] for i in range(F)])
        rx = RxPhy(cfg, agc=agc, variance_source=vs, max_batch=F)
        got = rx.receive(bb, taps=True)
        mabs = mrel = 0.0
        same = total = 0
        grid_same = True
        eq_rel = 0.0
        for f in range(F):
            ref = orc.rx(bb[f], flags)
            a, b = got["llr_ldpc"][f].astype(np.float64), ref["llr_ldpc"].astype(np.float64)
            d = np.abs(a - b)
            mabs = max(mabs, float(d.max()))
            mrel = max(mrel, float((d / np.maximum(1.0, np.abs(b))).max()))
            same += int((got["llr_ldpc"][f].view(np.uint32) == ref["llr_ldpc"].view(np.uint32)).sum())
            total += 1600
            grid_same &= got["grid"][f].tobytes() == ref["grid"].tobytes()
            eq_rel = max(eq_rel, float(np.abs(got["eq"][f] - ref["eq"]).max() / np.abs(ref["eq"]).max()))
        out[cfg] = {"max_abs_llr_err": mabs, "max_scaled_llr_err": mrel, "bit_identical_llrs": same / total,
                    "grid_bit_identical": bool(grid_same), "max_rel_eq_err": eq_rel}
        rx.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
