#!/bin/bash
# Round 4 A/B on a GPU box: kernel times of the product build and of the variant builds (mercury_amd/_variants/lib_*.so) at the operating
# point and on the headline workload.  tools/r04_ab.sh "<variants>" "<decoders>" "<esn0 list>" [cfg]
cd "$(dirname "$0")/.."
VARS=${1:-"base"}; DECS=${2:-"spa"}; ESS=${3:-"3.5 -15"}; CFG=${4:-8}
for rep in 1 2; do
for v in $VARS; do
  lib=""; [ "$v" != base ] && lib=$PWD/mercury_amd/_variants/lib_$v.so
  for d in $DECS; do for es in $ESS; do
    MERCURY_GPU_LIB=$lib python bench.py --cfg $CFG --decoder $d --esn0 $es --no-extras --no-cpu-baseline --steps 30 --warmup 3 $EXTRA 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-8s cfg %3d %-8s %6s dB: frontend %.4f ms  ldpc %.4f ms  %.0f frames/s  iters %.2f' % ('$v', $CFG, '$d', '$es', d['kernel_ms']['frontend'], d['kernel_ms']['ldpc'], d['value'], d['avg_iters_per_frame']))"
  done; done
done
done
