#!/bin/bash
# same-box A/B of library builds on the decoder's points: tools/ab_r06.sh "<variants: base = product build, else mercury_amd/_variants/lib_<name>.so>" ["<cfg:esn0[:variant] ...>"]
cd "$(dirname "$0")/.."
run() { # variant cfg esn0 extra
  lib=""; [ "$1" != base ] && lib=$PWD/mercury_amd/_variants/lib_$1.so
  MERCURY_GPU_LIB=$lib timeout 120 python bench.py --cfg $2 --esn0 $3 $4 --no-extras --no-cpu-baseline --steps 20 --warmup 3 --line compact 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-8s cfg %3d %6s dB %-24s fe %.4f ldpc %.4f ms  it %.2f  frac %.3f' % ('$1', $2, '$3', '$4', d['kernel_ms']['frontend'], d['kernel_ms']['ldpc'], d['avg_iters_per_frame'], d['roofline']['frac']))"
}
PTS=${2:-"8:-15 8:-1 8:3.5 16:-15:bbt 16:13:bbt 0:-15 11:-15"}
for rep in 1 2; do for v in ${1:-base}; do
  for p in $PTS; do IFS=: read cfg es var <<< "$p"; x=""; [ "$var" = bbt ] && x="--variant baseband_test"; run $v $cfg $es "$x"; done
done; done
