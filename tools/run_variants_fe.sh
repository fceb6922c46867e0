#!/bin/bash
# On a GPU box: front-end launch time with each library variant under mercury_amd/_variants (CFGS selects the modes)
cd "$(dirname "$0")/.."
cp mercury_amd/libmercury_gpu.so /tmp/lib_keep.so
for rep in 1 2; do
for lib in mercury_amd/_variants/lib_*.so; do
  cp $lib mercury_amd/libmercury_gpu.so
  for cfg in ${CFGS:-8}; do
    python bench.py --cfg $cfg --decoder spa_fast --esn0 3.5 --no-extras --no-cpu-baseline --steps 30 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-16s cfg %2d: frontend %.4f ms  ldpc %.4f' % ('$(basename $lib)', $cfg, d['kernel_ms']['frontend'], d['kernel_ms']['ldpc']))"
  done
done
done
cp /tmp/lib_keep.so mercury_amd/libmercury_gpu.so
