#!/bin/bash
# A/B of the fp64 sum-product kernel variants on the headline workload and on mode 16 (rate 14/16).
# usage: tools/ab_spa.sh [variants...]   (MERCURY_SPA_VARIANT values; default "0 1")
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=${@:-0 1}
for cfg in 8 16; do
  for v in $V; do
    MERCURY_SPA_VARIANT=$v python bench.py --cfg $cfg --no-extras --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/ab_cfg${cfg}_v${v}.json
    python - <<PY
import json
d=json.load(open("gpurun_out/ab_cfg${cfg}_v${v}.json"))
print("cfg ${cfg} variant ${v}: %.3f ms ldpc, %.0f frames/s, frac %.3f" % (d["kernel_ms"]["ldpc"], d["value"], d["roofline"]["frac"]))
PY
  done
done
