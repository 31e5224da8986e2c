export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r6_b_pytest_gpu.txt 2>&1; tail -3 gpurun_out/r6_b_pytest_gpu.txt
bash tools/gpu_round.sh r6_b > gpurun_out/r6_b_round.log 2>&1; tail -25 gpurun_out/r6_b_round.log
