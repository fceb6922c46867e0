#!/bin/bash
# Reproduces the evidence under profiles/ on a GPU box (run from the repo root; writes to gpurun_out/r02/, copy what you want judged
# into profiles/). Counter passes are separate rocprofv3 runs with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md
# prescribes; never combine --pmc with sys/hip/hsa traces on this pool.
#   tools/collect_profiles.sh            bench lines + kernel stats + PMC opcode mix / stall counters (cfg 8: spa, spa_fast, minsum) + opcode costs
#   tools/collect_profiles.sh sweep      additionally: decoder comparison on all 20 modes, the 20-mode throughput sweep, sync blocks,
#                                        receive_byte chain, host-buffer path, transmit chain
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r02
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
[ -x "$ROOT/tools/ubench/valu_cycles" ] || hipcc --offload-arch=gfx950 -O2 -o "$ROOT/tools/ubench/valu_cycles" "$ROOT/tools/ubench/valu_cycles.hip"
"$ROOT/tools/ubench/valu_cycles" > "$OUT/r02_valu_cycles.json"
[ -x "$ROOT/tools/ubench/dep_chain" ] || hipcc --offload-arch=gfx950 -O2 -o "$ROOT/tools/ubench/dep_chain" "$ROOT/tools/ubench/dep_chain.hip" 2>/dev/null
"$ROOT/tools/ubench/dep_chain" > "$OUT/r02_dep_chain.json"
for d in spa spa_fast minsum; do
  python "$ROOT/bench.py" --decoder $d > "$OUT/r02_bench_${d}_cfg8.json" 2>/dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$d" -- python "$ROOT/bench.py" --decoder $d --no-cpu-baseline --no-extras > /dev/null 2>&1
  cp "$(find "$OUT/prof_$d" -name '*kernel_stats.csv' | head -1)" "$OUT/r02_bench_${d}_cfg8_kernel_stats.csv"
  rm -rf "$OUT/prof_$d"
  "$ROOT/tools/collect_pmc_mix.sh" $d "$OUT/pmc_mix_$d.json" > /dev/null 2> "$OUT/pmc_mix_$d.err" || true
done
python - "$OUT" <<'PY'
import json, sys, os
out = sys.argv[1]
mix = {}
for d in ("spa", "spa_fast", "minsum"):
    f = os.path.join(out, "pmc_mix_%s.json" % d)
    if os.path.exists(f):
        for k, v in json.load(open(f)).items():
            if "ldpc" in k:
                mix[d] = dict(v, kernel=k)
            elif "frontend" in k:
                mix["frontend"] = dict(v, kernel=k)
json.dump(mix, open(os.path.join(out, "r02_instruction_mix.json"), "w"), indent=1)
PY
if [ "$1" = "sweep" ]; then
  cd "$ROOT"
  python tools/compare_decoders.py 4096 > "$OUT/r02_compare_decoders.json" 2> "$OUT/r02_compare_decoders.txt"
  python tools/sweep_modes.py > "$OUT/r02_mode_sweep.json" 2> "$OUT/r02_mode_sweep.txt"
  python tools/bench_sync.py > "$OUT/r02_bench_sync_blocks.json"
  python tests/tools/bench_receive_byte.py 8 1024 > "$OUT/r02_bench_receive_byte_cfg8.json"
  tools/timeline_receive_byte.sh 8 1024 > "$OUT/r02_receive_byte_timeline.txt" 2>/dev/null || true
  python tools/bench_host_path.py 8 4096 -15 > "$OUT/r02_bench_host_path_cfg8.json"
  python tools/bench_tx.py 8 4096 > "$OUT/r02_bench_tx_cfg8.json"
  python tools/bench_tsync_variants.py > "$OUT/r02_bench_tsync_variants.json" 2>/dev/null
  python tools/bench_tsync_fine.py > "$OUT/r02_bench_tsync_fine.json" 2>/dev/null
  tools/pmc_any.sh tsync_metric_fine "$OUT/r02_pmc_tsync_fine.json" -- python tools/bench_tsync_fine.py 1024 > /dev/null 2>&1 || true
  tools/pmc_any.sh p2b_slide_d1_kernel "$OUT/r02_pmc_p2b.json" -- python tools/bench_sync.py > /dev/null 2>&1 || true
  tools/pmc_any.sh tsync_metric_stream "$OUT/r02_pmc_tsync_stream.json" -- python tools/bench_tsync_variants.py 1024 > /dev/null 2>&1 || true
  tools/pmc_any.sh mfsk_frontend "$OUT/r02_pmc_mfsk_frontend.json" -- python bench.py --cfg 100 --decoder spa_fast --steps 3 --warmup 1 --no-cpu-baseline --no-extras --frames 2048 > /dev/null 2>&1 || true
  python bench.py --cfg 100 --decoder spa_fast --no-cpu-baseline --no-extras > "$OUT/r02_bench_spa_fast_cfg100.json" 2>/dev/null
  python bench.py --cfg 0 --decoder spa_fast --no-cpu-baseline --no-extras > "$OUT/r02_bench_spa_fast_cfg0.json" 2>/dev/null
  python bench.py --cfg 0 --decoder spa --no-cpu-baseline --no-extras > "$OUT/r02_bench_spa_cfg0.json" 2>/dev/null
fi
ls -la "$OUT"
