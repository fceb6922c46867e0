#!/bin/bash
# On a GPU box: decoder launch time at several Es/N0 with each library variant (DECS, ESN0S, CFGS select)
cd "$(dirname "$0")/.."
cp mercury_amd/libmercury_gpu.so /tmp/lib_keep.so
for lib in mercury_amd/_variants/lib_*.so; do
  cp $lib mercury_amd/libmercury_gpu.so
  for cfg in ${CFGS:-8}; do for dec in ${DECS:-spa_fast}; do for es in ${ESN0S:-3.5}; do
    python bench.py --cfg $cfg --decoder $dec --esn0 $es --no-extras --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-16s cfg %2d %-8s %5s dB: ldpc %.4f ms  iters %.2f' % ('$(basename $lib)', $cfg, '$dec', '$es', d['kernel_ms']['ldpc'], d['avg_iters_per_frame']))"
  done; done; done
done
cp /tmp/lib_keep.so mercury_amd/libmercury_gpu.so
