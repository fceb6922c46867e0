#!/usr/bin/env python3
"""profiles/hbm_traffic.json (read by bench.py for `roofline.traffic`) and profiles/r01_pmc_sq_summary.json from the rocprofv3
counter CSVs that tools/collect_profiles.sh leaves: HBM bytes per decoder-kernel launch = (2 * FETCH_SIZE + WRITE_SIZE) KB,
FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950's wide coalesced reads.
Usage: python tools/hbm_traffic_from_pmc.py [dir with pmc_*_cfg8.csv, default gpurun_out] [output dir, default profiles]"""
import collections
import csv
import json
import os
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
    dst = sys.argv[2] if len(sys.argv) > 2 else "profiles"
    out = {"_comment": "HBM bytes per kernel launch (4096 mode-8 frames, 50 iterations) from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                       "separate passes of `bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras` (tools/collect_profiles.sh). "
                       "FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 wide coalesced reads; raw values are KB."}
    summary = {}
    for d in ("spa", "minsum"):
        f = per_kernel(os.path.join(src, "pmc_FETCH_SIZE_%s_cfg8.csv" % d), "FETCH_SIZE")
        w = per_kernel(os.path.join(src, "pmc_WRITE_SIZE_%s_cfg8.csv" % d), "WRITE_SIZE")
        kern = [k for k in f if "ldpc" in k][0]
        fe = [k for k in f if "frontend" in k][0]
        out[d + "_cfg8"] = (2 * f[kern] + w[kern]) * 1024
        out[d + "_cfg8_detail"] = {"kernel": kern, "FETCH_SIZE_KB_raw": f[kern], "WRITE_SIZE_KB_raw": w[kern],
                                   "frontend_FETCH_SIZE_KB_raw": f[fe], "frontend_WRITE_SIZE_KB_raw": w[fe]}
        sq = os.path.join(src, "pmc_sq_%s_cfg8.csv" % d)
        if os.path.exists(sq):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(sq)):
                if r["Kernel_Name"] == kern:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            summary[d] = {"kernel": kern, **{c: sum(v) / len(v) for c, v in acc.items()}}
            s = summary[d]
            if "SQ_ACTIVE_INST_VALU" in s and "SQ_BUSY_CYCLES" in s:
                # per-SIMD VALU issue utilisation: active VALU cycles over (busy cycles x 4 SIMDs x CUs with waves); informative
                s["valu_active_cycles_per_busy_cycle"] = s["SQ_ACTIVE_INST_VALU"] / s["SQ_BUSY_CYCLES"]
    json.dump(out, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
    if summary:
        json.dump(summary, open(os.path.join(dst, "r01_pmc_sq_summary.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k.endswith("_cfg8")}))


if __name__ == "__main__":
    main()
