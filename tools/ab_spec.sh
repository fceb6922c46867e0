#!/bin/bash
# same-box A/B of the fp64 decoder's look policy: tools/ab_spec.sh "<weights: MERCURY_SPA_SPEC_WEIGHT values; 'old' = mercury_amd/_ab/lib_r6base.so>" ["<cfg:esn0[:bbt] ...>"]
cd "$(dirname "$0")/.."
run() { # weight cfg esn0 extra
  lib=""; w=$1; [ "$1" = old ] && { lib=$PWD/mercury_amd/_ab/lib_r6base.so; w=45; }
  MERCURY_SPA_SPEC_WEIGHT=$w MERCURY_GPU_LIB=$lib timeout 120 python bench.py --cfg $2 --esn0 $3 $4 --no-extras --no-cpu-baseline --steps 20 --warmup 3 --line compact 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-8s cfg %3d %6s dB %-24s fe %.4f ldpc %.4f ms  it %.2f  frac %.3f' % ('$1', $2, '$3', '$4', d['kernel_ms']['frontend'], d['kernel_ms']['ldpc'], d['avg_iters_per_frame'], d['roofline']['frac']))"
}
PTS=${2:-"8:-15 8:-1 8:1.5 8:2.5 8:3.5 8:6 4:-1.5 11:6 12:8.5 16:13:bbt"}
for rep in $(seq 1 ${REPS:-2}); do for v in ${1:-old 45}; do
  for p in $PTS; do IFS=: read cfg es var <<< "$p"; x=""; [ "$var" = bbt ] && x="--variant baseband_test"; run $v $cfg $es "$x"; done
done; done
