#!/bin/bash
# Reproduces the evidence under profiles/ on a GPU box (run from the repo root; writes to gpurun_out/$R/ with R = the round prefix,
# default r04; copy what you want judged into profiles/). Counter passes are separate rocprofv3 runs with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md
# prescribes; never combine --pmc with sys/hip/hsa traces on this pool.
#   tools/collect_profiles.sh            bench lines + kernel stats + PMC opcode mix / stall counters (cfg 8: spa, spa_fast, minsum) + opcode costs
#   tools/collect_profiles.sh sweep      additionally: decoder comparison on all 20 modes, the 20-mode throughput sweep, sync blocks,
#                                        receive_byte chain, host-buffer path, transmit chain
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
R=${R:-r06}
OUT=$ROOT/gpurun_out/$R
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# bench.py prints the full record and then the compact contract record (the last line): <name>_full.json and <name>.json
bench_pair() { local name=$1; shift; python "$ROOT/bench.py" "$@" 2>/dev/null > "$OUT/.pair"; head -1 "$OUT/.pair" > "$OUT/${name}_full.json"; tail -1 "$OUT/.pair" > "$OUT/${name}.json"; rm -f "$OUT/.pair"; }
[ -x "$ROOT/tools/ubench/valu_cycles" ] || hipcc --offload-arch=gfx950 -O2 -o "$ROOT/tools/ubench/valu_cycles" "$ROOT/tools/ubench/valu_cycles.hip"
"$ROOT/tools/ubench/valu_cycles" > "$OUT/${R}_valu_cycles.json"
cp "$OUT/${R}_valu_cycles.json" "$ROOT/profiles/${R}_valu_cycles.json"       # where bench.py looks for the opcode costs
[ -x "$ROOT/tools/ubench/dep_chain" ] || hipcc --offload-arch=gfx950 -O2 -o "$ROOT/tools/ubench/dep_chain" "$ROOT/tools/ubench/dep_chain.hip" 2>/dev/null
"$ROOT/tools/ubench/dep_chain" > "$OUT/${R}_dep_chain.json"
[ -x "$ROOT/tools/ubench/lds_mask" ] || hipcc --offload-arch=gfx950 -O2 -o "$ROOT/tools/ubench/lds_mask" "$ROOT/tools/ubench/lds_mask.hip" 2>/dev/null
"$ROOT/tools/ubench/lds_mask" > "$OUT/${R}_lds_mask.json"            # LDS-pipeline cycles per instruction kind / execution mask / address pattern (NOTES R5.9)
for d in spa spa_fast minsum; do
  "$ROOT/tools/collect_pmc_mix.sh" $d "$OUT/pmc_mix_$d.json" > /dev/null 2> "$OUT/pmc_mix_$d.err" || true
done
"$ROOT/tools/collect_pmc_mix.sh" spa "$OUT/pmc_mix_spa_cfg16.json" --cfg 16 --variant baseband_test > /dev/null 2> "$OUT/pmc_mix_spa_cfg16.err" || true
# the operating-point launch (mode 8 at Es/N0 3.5 dB, 3.75 iterations per frame: SURVEY.md 8d C2's second point) gets PMC passes of its own
OPES=${OPES:-3.5}
for d in spa spa_fast minsum; do
  "$ROOT/tools/collect_pmc_mix.sh" $d "$OUT/pmc_mix_${d}_op.json" --esn0 $OPES > /dev/null 2> "$OUT/pmc_mix_${d}_op.err" || true
done
# the waterfall launch (mode 8 just below its threshold: every frame runs all 50 iterations on LLRs of real magnitude) too
WFES=${WFES:--1.0}
"$ROOT/tools/collect_pmc_mix.sh" spa "$OUT/pmc_mix_spa_wf.json" --esn0 $WFES > /dev/null 2> "$OUT/pmc_mix_spa_wf.err" || true
"$ROOT/tools/collect_pmc_mix.sh" spa "$OUT/pmc_mix_spa_cfg16_13db.json" --cfg 16 --variant baseband_test --esn0 13 > /dev/null 2> "$OUT/pmc_mix_spa_cfg16_13db.err" || true
python - "$OUT" "$ROOT" "$R" <<'PY'
import json, sys, os
out, root, R = sys.argv[1:4]
sys.path.insert(0, root)
from mercury_amd.build import decoder_digest
mix = {"decoder_digest": decoder_digest(),
       "note": "average per launch of the headline workload (bench.py defaults: 4096 mode-8 frames, 50 iterations each; keys ending in _op: the same frames count at Es/N0 3.5 dB, 3.75 iterations per frame); separate rocprofv3 --pmc "
               "passes (tools/collect_pmc_mix.sh); bench.py quotes this file only while decoder_digest matches mercury_amd.build.decoder_digest()"}
for d in ("spa", "spa_fast", "minsum"):
    f = os.path.join(out, "pmc_mix_%s.json" % d)
    if os.path.exists(f):
        for k, v in json.load(open(f)).items():
            if "ldpc" in k:
                mix[d] = dict(v, kernel=k)
            elif "frontend" in k:
                mix["frontend"] = dict(v, kernel=k)
for d in ("spa", "spa_fast", "minsum"):
    f = os.path.join(out, "pmc_mix_%s_op.json" % d)
    if os.path.exists(f):
        for k, v in json.load(open(f)).items():
            if "ldpc" in k:
                mix[d + "_op"] = dict(v, kernel=k, workload="mode 8 at Es/N0 3.5 dB (operating point)")
            elif "frontend" in k:
                mix["frontend_op"] = dict(v, kernel=k)
for key, name, wl in (("spa_cfg16", "pmc_mix_spa_cfg16.json", "mode 16 baseband_test at -15 dB"), ("spa_cfg16_13db", "pmc_mix_spa_cfg16_13db.json", "mode 16 baseband_test at 13 dB"),
                      ("spa_wf", "pmc_mix_spa_wf.json", "mode 8 just below its threshold (waterfall_point)")):
    f = os.path.join(out, name)
    if os.path.exists(f):
        for k, v in json.load(open(f)).items():
            if "ldpc" in k:
                mix[key] = dict(v, kernel=k, workload=wl)
            elif "frontend" in k and key == "spa_wf":
                mix["frontend_wf"] = dict(v, kernel=k)
json.dump(mix, open(os.path.join(out, "%s_instruction_mix.json" % R), "w"), indent=1)
json.dump(mix, open(os.path.join(root, "profiles", "%s_instruction_mix.json" % R), "w"), indent=1)     # where bench.py looks for it
PY
for d in spa spa_fast minsum; do
  bench_pair ${R}_bench_${d}_cfg8 --decoder $d
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$d" -- python "$ROOT/bench.py" --decoder $d --no-cpu-baseline --no-extras > /dev/null 2>&1
  cp "$(find "$OUT/prof_$d" -name '*kernel_stats.csv' | head -1)" "$OUT/${R}_bench_${d}_cfg8_kernel_stats.csv"
  rm -rf "$OUT/prof_$d"
done
# the operating point: kernel stats of the launch bench.py reports as `operating_point`, the front-end's phase stamps, the decoders' phase stamps
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_op" -- python "$ROOT/bench.py" --esn0 $OPES --no-cpu-baseline --no-extras --steps 30 > /dev/null 2>&1
cp "$(find "$OUT/prof_op" -name '*kernel_stats.csv' | head -1)" "$OUT/${R}_op_spa_cfg8_kernel_stats.csv" 2>/dev/null; rm -rf "$OUT/prof_op"
( cd "$ROOT" && python tools/fe_phases.py 8 4096 $OPES > "$OUT/${R}_op_frontend_phases_cfg8.txt" 2>/dev/null; python tools/fe_phases.py 0 4096 -7 > "$OUT/${R}_op_frontend_phases_cfg0.txt" 2>/dev/null
  if [ -f mercury_amd/_variants/lib_stamps.so ]; then
    for spec in "spa $OPES" "spa_fast $OPES" "spa -15"; do set -- $spec
      MERCURY_GPU_LIB=$ROOT/mercury_amd/_variants/lib_stamps.so python tools/spa_stamps.py $1 8 4096 $2 2>/dev/null | grep -v "^{" > "$OUT/${R}_decoder_phases_${1}_es${2}.txt"
    done
  fi )
# the waterfall point: kernel stats of the launch bench.py reports as `waterfall_point`
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_wf" -- python "$ROOT/bench.py" --esn0 $WFES --no-cpu-baseline --no-extras --steps 30 > /dev/null 2>&1
cp "$(find "$OUT/prof_wf" -name '*kernel_stats.csv' | head -1)" "$OUT/${R}_wf_spa_cfg8_kernel_stats.csv" 2>/dev/null; rm -rf "$OUT/prof_wf"
# the high-degree graph (rate 14/16: modes 12, 14, 15, 16; BASELINE.json configs[3]), in noise and at 13 dB (50 iterations both, LLRs of real magnitude at 13 dB)
python "$ROOT/bench.py" --cfg 16 --variant baseband_test --no-extras --line compact > "$OUT/${R}_bench_spa_cfg16.json" 2>/dev/null
python "$ROOT/bench.py" --cfg 16 --variant baseband_test --esn0 13 --no-extras --no-cpu-baseline --line compact > "$OUT/${R}_bench_spa_cfg16_13db.json" 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_16b" -- python "$ROOT/bench.py" --cfg 16 --variant baseband_test --esn0 13 --no-cpu-baseline --no-extras > /dev/null 2>&1
cp "$(find "$OUT/prof_16b" -name '*kernel_stats.csv' | head -1)" "$OUT/${R}_bench_spa_cfg16_13db_kernel_stats.csv" 2>/dev/null; rm -rf "$OUT/prof_16b"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_16" -- python "$ROOT/bench.py" --cfg 16 --variant baseband_test --no-cpu-baseline --no-extras > /dev/null 2>&1
cp "$(find "$OUT/prof_16" -name '*kernel_stats.csv' | head -1)" "$OUT/${R}_bench_spa_cfg16_kernel_stats.csv"
rm -rf "$OUT/prof_16"
python "$ROOT/bench.py" --cfg 16 --variant baseband_test --decoder spa_fast --no-cpu-baseline --no-extras --line compact > "$OUT/${R}_bench_spa_fast_cfg16.json" 2>/dev/null
python "$ROOT/bench.py" --gpus 1 --pool --no-extras --line compact > "$OUT/${R}_bench_spa_cfg8_pool.json" 2>/dev/null
if [ "$1" = "sweep" ]; then
  cd "$ROOT"
  python tools/compare_decoders.py 4096 > "$OUT/${R}_compare_decoders.json" 2> "$OUT/${R}_compare_decoders.txt"
  python tools/sweep_modes.py > "$OUT/${R}_mode_sweep.json" 2> "$OUT/${R}_mode_sweep.txt"
  python tools/bench_sync.py > "$OUT/${R}_bench_sync_blocks.json"
  python tests/tools/bench_receive_byte.py 8 1024 > "$OUT/${R}_bench_receive_byte_cfg8.json"
  tools/timeline_receive_byte.sh 8 1024 > "$OUT/${R}_receive_byte_timeline.txt" 2>/dev/null || true
  python tools/bench_host_path.py 8 4096 -15 > "$OUT/${R}_bench_host_path_cfg8.json"
  python tools/bench_tx.py 8 4096 > "$OUT/${R}_bench_tx_cfg8.json"
  python tools/bench_tsync_variants.py > "$OUT/${R}_bench_tsync_variants.json" 2>/dev/null
  python tools/bench_tsync_fine.py > "$OUT/${R}_bench_tsync_fine.json" 2>/dev/null
  tools/pmc_any.sh tsync_metric_fine "$OUT/${R}_pmc_tsync_fine.json" -- python tools/bench_tsync_fine.py 1024 > /dev/null 2>&1 || true
  tools/pmc_any.sh p2b_slide_d1_kernel "$OUT/${R}_pmc_p2b.json" -- python tools/bench_sync.py > /dev/null 2>&1 || true
  tools/pmc_any.sh tsync_metric_stream "$OUT/${R}_pmc_tsync_stream.json" -- python tools/bench_tsync_variants.py 1024 > /dev/null 2>&1 || true
  tools/pmc_any.sh mfsk_frontend "$OUT/${R}_pmc_mfsk_frontend.json" -- python bench.py --cfg 100 --decoder spa_fast --steps 3 --warmup 1 --no-cpu-baseline --no-extras --frames 2048 > /dev/null 2>&1 || true
  python bench.py --cfg 100 --decoder spa_fast --no-cpu-baseline --no-extras --line compact > "$OUT/${R}_bench_spa_fast_cfg100.json" 2>/dev/null
  python bench.py --cfg 0 --decoder spa_fast --no-cpu-baseline --no-extras --line compact > "$OUT/${R}_bench_spa_fast_cfg0.json" 2>/dev/null
  python bench.py --cfg 0 --decoder spa --no-cpu-baseline --no-extras --line compact > "$OUT/${R}_bench_spa_cfg0.json" 2>/dev/null
  # BASELINE.json configs[4] on one GPU: decoder-only, rate 8/16 (mode 13's code), noise-only LLRs so that every codeword runs all its iterations
  : > "$OUT/${R}_bench_ldpc_only_rate8.jsonl"
  for dec in spa spa_fast minsum; do for it in 5 20 50; do
    python bench.py --ldpc-only --cfg 13 --iters $it --decoder $dec --no-cpu-baseline --no-extras --steps 20 2>/dev/null | tail -1 >> "$OUT/${R}_bench_ldpc_only_rate8.jsonl"
  done; done
fi
# which of fdlibm's case branches the decoder's wavefronts enter (variant build: tools/build_variants.sh census:"-DSPA_CENSUS_ON=1")
if [ -f "$ROOT/mercury_amd/_variants/lib_census.so" ]; then
  ( cd "$ROOT"; for a in "8 1024 -15" "8 1024 -1" "8 1024 3.5" "16 512 -15" "16 512 -15 baseband" "14 512 -15" "16 512 13 baseband" "0 512 -15" "11 512 -15"; do
      MERCURY_GPU_LIB=$ROOT/mercury_amd/_variants/lib_census.so python tools/spa_census.py $a 2>/dev/null | tail -25; done > "$OUT/${R}_spa_branch_census.txt" )
fi
ls -la "$OUT"
