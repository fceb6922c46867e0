#!/bin/bash
# Same-box A/B of library builds: tools/ab_lib.sh "<variants: base = the product build, else mercury_amd/_variants/lib_<name>.so>" "<esn0 list>" "<cfgs>"
cd "$(dirname "$0")/.."
for rep in 1 2; do for v in ${1:-base}; do
  lib=""; [ "$v" != base ] && lib=$PWD/mercury_amd/_variants/lib_$v.so
  for cfg in ${3:-8}; do for es in ${2:--15}; do
    MERCURY_GPU_LIB=$lib timeout 120 python bench.py --cfg $cfg --esn0 $es --no-extras --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-6s cfg %3d %6s dB: frontend %.4f ms  ldpc %.4f ms' % ('$v', $cfg, '$es', d['kernel_ms']['frontend'], d['kernel_ms']['ldpc']))"
  done; done
done; done
