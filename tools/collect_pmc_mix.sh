#!/bin/bash
# Dynamic instruction mix + stall counters of the decoder kernel of one bench.py workload (run on a GPU box, from anywhere).
#   tools/collect_pmc_mix.sh <decoder: spa|minsum|spa_fast> [out.json] [extra bench.py args...]
# One rocprofv3 pass per counter group, --kernel-trace only (never combined with sys/hip/hsa traces on this pool).
# Output: JSON {kernel: {counter: average per launch}} for the decoder and front-end kernels.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
DEC=${1:-spa}; OUT=${2:-$ROOT/gpurun_out/pmc_mix_$DEC.json}; shift 2 || true
OUT=$(realpath -m "$OUT"); mkdir -p "$(dirname "$OUT")"
TMP=$(mktemp -d /tmp/pmcmix.XXXX)
cd /tmp && export TMPDIR=/tmp
GRPS=(
 "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64"
 "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
 "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32"
 "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM"
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY"
 "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
 "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"
)
i=0
for grp in "${GRPS[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $TMP/g$i -- python "$ROOT/bench.py" --decoder $DEC --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > /dev/null 2> $TMP/g$i.err || echo "group $i failed: $grp" >&2
done
python - "$TMP" "$OUT" <<'PY'
import csv, sys, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "ldpc" in k or "frontend" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in acc.items()}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $TMP
