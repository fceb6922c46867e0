#!/bin/bash
# On a GPU box: times the headline decoder launch (and mode 16 when CFGS="8 16") with each library variant.
cd "$(dirname "$0")/.."
cp mercury_amd/libmercury_gpu.so /tmp/lib_keep.so
for rep in 1 2; do
for lib in mercury_amd/_variants/lib_*.so; do
  cp $lib mercury_amd/libmercury_gpu.so
  for cfg in ${CFGS:-8}; do
    python bench.py --cfg $cfg --no-extras --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-28s cfg %2d: %.3f ms' % ('$(basename $lib)', $cfg, d['kernel_ms']['ldpc']))"
  done
done
done
cp /tmp/lib_keep.so mercury_amd/libmercury_gpu.so
