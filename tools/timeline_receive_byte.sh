#!/bin/bash
# Kernel timeline of the last of a few device-resident receive_byte calls (start / duration / idle gap before, microseconds): where the
# host's control rounds leave the GPU idle.   usage (on a GPU box, from the repo root): tools/timeline_receive_byte.sh [cfg] [W] > out.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
D=$(mktemp -d /tmp/tl.XXXX)
rocprofv3 --kernel-trace --output-format csv -d $D -- python "$ROOT/tools/profile_receive_byte.py" ${1:-8} ${2:-1024} 3 2>&1 | grep ms_per_call
python - "$(find $D -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "window_energy" in r["Kernel_Name"]][-1]
j = idx
while j > 0 and "ldpc" not in rows[j]["Kernel_Name"]:
    j -= 1
rows = rows[j + 1:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
idle = 0.0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3
    if gap > 0:
        idle += gap
    print("%-40s start %8.1f dur %8.1f gap %7.1f" % (r["Kernel_Name"][:40], (s - t0) / 1e3, (e - s) / 1e3, gap))
    prev_end = max(prev_end, e)
print("total %.1f us, idle %.1f us" % ((prev_end - t0) / 1e3, idle))
PY
rm -rf $D
