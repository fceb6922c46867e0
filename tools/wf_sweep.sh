mkdir -p gpurun_out
for es in -15 -4 -3 -2 -1.5 -1 -0.5 0 0.5 1 2.5 3.5; do
  timeout 120 python bench.py --cfg 8 --esn0 $es --no-extras --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg 8 %6s dB: fe %.4f ldpc %.4f ms  avg_iters %.2f decoded %.3f frac %.3f' % ('$es', d['kernel_ms']['frontend'], d['kernel_ms']['ldpc'], d['avg_iters_per_frame'], d['decoded_fraction'], d['roofline']['frac']))"
done | tee gpurun_out/r06_wf_sweep.txt
for es in -15 8 13; do
  timeout 120 python bench.py --cfg 16 --variant baseband_test --esn0 $es --no-extras --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg 16 bbt %6s dB: fe %.4f ldpc %.4f ms  avg_iters %.2f decoded %.3f frac %.3f' % ('$es', d['kernel_ms']['frontend'], d['kernel_ms']['ldpc'], d['avg_iters_per_frame'], d['decoded_fraction'], d['roofline']['frac']))"
done | tee -a gpurun_out/r06_wf_sweep.txt
