cd /root/repo
for v in base fe7 base fe7; do
  lib=""; [ "$v" != base ] && lib=$PWD/mercury_amd/_variants/lib_$v.so
  MERCURY_GPU_LIB=$lib python bench.py --decoder spa_fast --esn0 3.5 --no-extras --no-cpu-baseline --steps 30 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v frontend %.4f ms' % (d['kernel_ms']['frontend']))"
done
for v in base fe7; do
  lib=""; [ "$v" != base ] && lib=$PWD/mercury_amd/_variants/lib_$v.so
  echo "== $v"; MERCURY_GPU_LIB=$lib python tools/fe_phases.py 8 4096 3.5 2>/dev/null
  MERCURY_GPU_LIB=$lib python -c "
import sys; sys.path.insert(0,'.')
from mercury_amd import RxPhy
rx=RxPhy(8,max_batch=8)
print('occupancy calc', rx.lib.mgpu_debug_occupancy(rx.h,0), 'lds', rx.lib.mgpu_debug_occupancy(rx.h,1))"
done
