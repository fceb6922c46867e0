#!/bin/bash
# Round 4: what the two kernels of the RX path do on the OPERATING-POINT launch (mode 8 at Es/N0 3.5 dB: 3.75 LDPC iterations per frame on average) —
# bench lines, rocprofv3 kernel stats, PMC opcode mix / activity counters (separate --pmc passes), phase stamps of the front-end (FE_STAMP) and of the
# decoders (SPA_STAMP variant build). Run on a GPU box from the repo root; writes gpurun_out/r04_op/.
#   tools/r04_op_profile.sh [tag]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-base}
OUT=$ROOT/gpurun_out/r04_op_$TAG
mkdir -p "$OUT"
cd "$ROOT"
ES=${ES:-3.5}
for d in spa spa_fast; do
  python bench.py --decoder $d --esn0 $ES --no-cpu-baseline --no-extras --steps 30 2>/dev/null | tail -1 > "$OUT/bench_${d}_op.json"
  python - "$OUT/bench_${d}_op.json" $d <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%-8s op: %.0f frames/s, kernels %s, %.2f iterations" % (sys.argv[2], d["value"], d["kernel_ms"], d["avg_iters_per_frame"]))
PY
done
python bench.py --cfg 0 --decoder spa_fast --esn0 -7 --no-cpu-baseline --no-extras --steps 30 2>/dev/null | tail -1 > "$OUT/bench_spa_fast_cfg0_op.json"
python -c "import json; d=json.load(open('$OUT/bench_spa_fast_cfg0_op.json')); print('cfg0 spa_fast op:', d['kernel_ms'])"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python "$ROOT/bench.py" --esn0 $ES --no-cpu-baseline --no-extras --steps 30 > /dev/null 2>&1 )
cp "$(find "$OUT/prof" -name '*kernel_stats.csv' | head -1)" "$OUT/bench_spa_op_kernel_stats.csv" 2>/dev/null; rm -rf "$OUT/prof"
for d in spa spa_fast; do
  tools/collect_pmc_mix.sh $d "$OUT/pmc_mix_${d}_op.json" --esn0 $ES > /dev/null 2> "$OUT/pmc_mix_${d}_op.err" || true
done
python tools/fe_phases.py 8 4096 $ES > "$OUT/fe_phases_cfg8.txt" 2>&1
python tools/fe_phases.py 0 4096 -7 > "$OUT/fe_phases_cfg0.txt" 2>&1
if [ -f mercury_amd/_variants/lib_stamps.so ]; then
  for spec in "spa $ES" "spa_fast $ES" "spa -15"; do
    set -- $spec
    MERCURY_GPU_LIB=$ROOT/mercury_amd/_variants/lib_stamps.so python tools/spa_stamps.py $1 8 4096 $2 > "$OUT/stamps_${1}_es${2}.txt" 2>&1
  done
fi
tail -n 14 "$OUT"/fe_phases_cfg8.txt "$OUT"/stamps_spa_es$ES.txt | cut -c1-200
