#!/bin/bash
# scratch: decoder timing at several Es/N0 (fixed cost at 30 dB = 0 iterations, operating point, worst case)
cd "$(dirname "$0")/.."
B="python bench.py --no-cpu-baseline --no-extras --steps 30"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), d['kernel_ms'], d.get('avg_iters_per_frame'))" "$1"; }
for dec in spa spa_fast minsum; do
  for es in 30 3.5 -15; do $B --decoder $dec --esn0 $es 2>/dev/null | pick "$dec esn0=$es"; done
done
for dec in spa_fast minsum; do $B --cfg 16 --variant baseband_test --decoder $dec 2>/dev/null | pick "$dec cfg16"; $B --cfg 0 --decoder $dec 2>/dev/null | pick "$dec cfg0"; done
