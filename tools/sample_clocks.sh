#!/bin/bash
# Samples sclk / power while the headline bench runs (is the decoder power-limited?). Run on a GPU box.
cd "$(dirname "$0")/.."
( python bench.py --no-extras --no-cpu-baseline --steps 2500 --warmup 3 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench ldpc ms', d['kernel_ms']['ldpc'])" ) &
BP=$!
for i in $(seq 1 16); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | sed 's/.*sclk clock level: S: //; s/.*Power (W): /W=/' | tr '\n' ' '; echo
  sleep 1
done
wait $BP
