#!/bin/bash
# On a GPU box: the fp64 decoder's launch time for 256 / 512 / 768 / 1024 codewords (one workgroup per codeword, 256 CUs): how the time
# grows from one workgroup per CU to two tells how many a CU really holds (2: x1.6 from 256 to 512; 1: x2). Modes 16 (rate 14/16) and 8.
cd "$(dirname "$0")/.."
B="python bench.py --no-cpu-baseline --no-extras --steps 20 --variant baseband_test"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['kernel_ms']['ldpc'])" "$1"; }
for cfg in 16 8; do for fr in 256 512 768 1024; do $B --cfg $cfg --frames $fr 2>/dev/null | pick "cfg$cfg frames$fr"; done; done
