#!/bin/bash
# Reproduces the evidence under profiles/ on a GPU box (run from the repo root; writes to gpurun_out/, copy what you
# want judged into profiles/). Counter passes are separate rocprofv3 runs with --kernel-trace only, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes; never combine --pmc with sys/hip/hsa traces on this pool.
#   tools/collect_profiles.sh            bench lines + kernel stats + PMC (cfg 8, spa and minsum)
#   tools/collect_profiles.sh sweep      additionally the 20-mode sweep, sync blocks, receive_byte chain
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for d in spa minsum; do
  python "$ROOT/bench.py" --decoder $d > "$OUT/bench_${d}_cfg8.json" 2>/dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$d" -- python "$ROOT/bench.py" --decoder $d --no-cpu-baseline --no-extras > /dev/null 2>&1
  cp "$(find "$OUT/prof_$d" -name '*kernel_stats.csv' | head -1)" "$OUT/bench_${d}_cfg8_kernel_stats.csv"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/pmc_${ctr}_$d" -- python "$ROOT/bench.py" --decoder $d --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
    cp "$(find "$OUT/pmc_${ctr}_$d" -name '*counter_collection.csv' | head -1)" "$OUT/pmc_${ctr}_${d}_cfg8.csv"
  done
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv \
    -d "$OUT/pmc_sq_$d" -- python "$ROOT/bench.py" --decoder $d --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1 || true
  f=$(find "$OUT/pmc_sq_$d" -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/pmc_sq_${d}_cfg8.csv"
done
if [ "$1" = "sweep" ]; then
  cd "$ROOT"
  python tools/sweep_modes.py > "$OUT/mode_sweep.json" 2> "$OUT/mode_sweep.txt"
  python tools/bench_sync.py > "$OUT/bench_sync_blocks.json"
  python tests/tools/bench_receive_byte.py 8 1024 > "$OUT/bench_receive_byte_cfg8.json"
  python tools/bench_tx.py 8 4096 > "$OUT/bench_tx_cfg8.json"
  python tools/bench_tx.py 100 512 > "$OUT/bench_tx_cfg100.json"
  python tests/tools/llr_error_table.py > "$OUT/llr_error_by_mode.json" 2>/dev/null || true
fi
python "$ROOT/tools/hbm_traffic_from_pmc.py" "$OUT" "$OUT" || true     # -> $OUT/hbm_traffic.json, r01_pmc_sq_summary.json
ls -la "$OUT"
