# round 6, fourth GPU call: C4 modes with the function tolerance step taken (default), pg tests
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python tools/dbg/c4_async.py > gpurun_out/r6d_c4_async.txt 2>&1; tail -14 gpurun_out/r6d_c4_async.txt
timeout 900 python -m pytest tests/test_gpu_pose_graph.py tests/test_gpu_fuzz_pose_graph.py -m gpu -q > gpurun_out/r6d_pg_tests.log 2>&1; tail -8 gpurun_out/r6d_pg_tests.log
