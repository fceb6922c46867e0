cd /root/repo
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fast or minsum or fp32 or mfsk" 2>&1 | tail -3
tools/r04_ab.sh "base old" "spa_fast minsum" "-15 3.5" 2>&1
tools/r04_ab.sh "base old" "spa_fast" "-15 20" 16 2>&1
tools/r04_ab.sh "base old" "spa_fast" "-15 -7" 0 2>&1 | tail -4
tools/pmc_any.sh ldpc gpurun_out/pmc_lds_new.json -- python bench.py --decoder spa_fast --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
python -c "
import json; d=json.load(open('gpurun_out/pmc_lds_new.json'))
for k,v in d.items(): print(k, {c:round(v[c]) for c in ('SQ_INSTS_LDS','SQ_LDS_BANK_CONFLICT','SQ_LDS_IDX_ACTIVE','SQ_ACTIVE_INST_LDS','SQ_WAIT_INST_LDS') if c in v})"
