cd /root/repo
B="python bench.py --no-cpu-baseline --no-extras --steps 20 --variant baseband_test"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['kernel_ms']['ldpc'])" "$1"; }
for cfg in 16 8; do for fr in 256 512 768 1024; do $B --cfg $cfg --frames $fr 2>/dev/null | pick "cfg$cfg frames$fr"; done; done
