export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python tools/dbg/lh_slices.py > gpurun_out/r6g_lh_slices.txt 2>&1; tail -12 gpurun_out/r6g_lh_slices.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_sharding.py tests/test_gpu_bad_inputs.py -m gpu -q -x > gpurun_out/r6g_tests.log 2>&1; tail -6 gpurun_out/r6g_tests.log
