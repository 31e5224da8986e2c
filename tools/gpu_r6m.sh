export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/drop_in_time.py > gpurun_out/r6m_dropin.json 2> gpurun_out/r6m_dropin.err; head -c 900 gpurun_out/r6m_dropin.json; echo
timeout 900 python -m pytest tests/test_cpp_shim.py tests/test_gpu_fuzz_dense.py -m gpu -q > gpurun_out/r6m_tests.log 2>&1; tail -2 gpurun_out/r6m_tests.log
(echo "# python tools/soak.py 4 on the final build of round 6 (MI355X)"; timeout 1200 python tools/soak.py 4 2>&1 | grep -v amdgpu) > gpurun_out/r6_soak.txt; cat gpurun_out/r6_soak.txt
