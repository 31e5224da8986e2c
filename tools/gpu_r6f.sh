# round 6: C4 default (adaptive lag), pose-graph tests + fuzz, then the whole GPU suite
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python tools/dbg/c4_async.py > gpurun_out/r6f_c4_async.txt 2>&1; tail -10 gpurun_out/r6f_c4_async.txt
timeout 900 python -m pytest tests/test_gpu_pose_graph.py tests/test_gpu_fuzz_pose_graph.py tests/test_sharding.py -m gpu -q > gpurun_out/r6f_pg_tests.log 2>&1; tail -8 gpurun_out/r6f_pg_tests.log
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_pose_graph.py --deselect tests/test_gpu_fuzz_pose_graph.py --deselect tests/test_sharding.py > gpurun_out/r6f_pytest.log 2>&1; tail -5 gpurun_out/r6f_pytest.log
