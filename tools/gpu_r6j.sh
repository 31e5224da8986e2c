export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python tools/dbg/c4_async.py > gpurun_out/r6j_c4_async.txt 2>&1; tail -9 gpurun_out/r6j_c4_async.txt
timeout 900 python -m pytest tests/test_gpu_pose_graph.py tests/test_gpu_fuzz_pose_graph.py tests/test_sharding.py tests/test_cpp_shim.py -m gpu -q > gpurun_out/r6j_pg_tests.log 2>&1; tail -3 gpurun_out/r6j_pg_tests.log
bash tools/gpu_c4_trace.sh > gpurun_out/r6j_c4_iter_trace.txt 2>&1; grep -v "chol_\|fillBuffer" gpurun_out/r6j_c4_iter_trace.txt | tail -40
