# round 6, first GPU call: the drop-in path's wall-clock (PnP published workload, C5 through ceres::Solve) + the shim tests
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_cpp_shim.py -m gpu -x -q > gpurun_out/r6a_shim.log 2>&1; tail -5 gpurun_out/r6a_shim.log
timeout 900 python tools/drop_in_time.py --c5 > gpurun_out/r6a_dropin.json 2> gpurun_out/r6a_dropin.err; cat gpurun_out/r6a_dropin.json | cut -c1-3000; tail -3 gpurun_out/r6a_dropin.err
timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_cpp_shim.py > gpurun_out/r6a_pytest.log 2>&1; tail -5 gpurun_out/r6a_pytest.log
