export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_cpp_shim.py tests/test_gpu_fuzz_dense.py -m gpu -q -x > gpurun_out/r6l_tests.log 2>&1; tail -4 gpurun_out/r6l_tests.log
timeout 600 python tools/drop_in_time.py > gpurun_out/r6l_dropin.json 2> gpurun_out/r6l_dropin.err; head -c 1500 gpurun_out/r6l_dropin.json; tail -2 gpurun_out/r6l_dropin.err
