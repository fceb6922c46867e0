#!/bin/bash
# The fp64 sum-product decoder launch on the headline workload (mode 8, rate 6/16) and on mode 16 (rate 14/16): ms per 4096 x 50.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for cfg in ${@:-8 16}; do
  python bench.py --cfg $cfg --no-extras --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/spa_cfg${cfg}.json
  python - <<PY
import json
d=json.load(open("gpurun_out/spa_cfg${cfg}.json"))
print("cfg ${cfg}: %.3f ms ldpc, %.0f frames/s, frac %.3f" % (d["kernel_ms"]["ldpc"], d["value"], d["roofline"]["frac"]))
PY
done
