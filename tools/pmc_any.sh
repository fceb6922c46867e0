#!/bin/bash
# Stall / activity counters of the kernels matching a name pattern for an arbitrary command (run on a GPU box).
#   tools/pmc_any.sh <kernel-name-substring> <out.json> -- <command...>
# One rocprofv3 pass per counter group, --kernel-trace only.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PAT=$1; OUT=$(realpath -m "$2"); shift 3
mkdir -p "$(dirname "$OUT")"
TMP=$(mktemp -d /tmp/pmcany.XXXX)
CMD=("$@")
cd /tmp && export TMPDIR=/tmp
GRPS=(
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM"
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY"
 "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
 "SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32"
 "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"
)
i=0
for grp in "${GRPS[@]}"; do
  i=$((i+1))
  (cd "$ROOT" && rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $TMP/g$i -- "${CMD[@]}" > /dev/null 2> $TMP/g$i.err) || echo "group $i failed: $grp" >&2
done
python - "$TMP" "$OUT" "$PAT" <<'PY'
import csv, sys, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if sys.argv[3] in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in acc.items()}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $TMP
