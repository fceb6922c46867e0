#!/bin/bash
# scratch: per-kernel times of the RX path at several operating points (bench.py, no CPU baseline, no extras)
cd "$(dirname "$0")/.."
B="python bench.py --no-cpu-baseline --no-extras --steps 30"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), d['kernel_ms'], d.get('avg_iters_per_frame'))" "$1"; }
for dec in ${DECS:-spa spa_fast minsum}; do
  for es in ${ESN0S:-30 3.5 -15}; do $B --decoder $dec --esn0 $es 2>/dev/null | pick "$dec esn0=$es"; done
done
