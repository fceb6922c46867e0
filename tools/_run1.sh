cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/spec_parity4.txt 2>&1
tail -15 gpurun_out/spec_parity4.txt
tools/ab_spec.sh "old 100,45" "8:-15 8:-1 8:2.5 8:3.5 16:-15:bbt 16:13:bbt 12:8.5 14:11 0:-8 4:-1.5" > gpurun_out/ab_spec7.txt 2>&1
