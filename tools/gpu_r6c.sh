# round 6, third GPU call: C4 with the coarse inverse on its second stream (consistent tree), host CPU limits, thread sweep with binding
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)  affinity: $(taskset -p $$ 2>/dev/null | tail -1)" > gpurun_out/r6c_host.txt
lscpu | grep -E "Model name|Socket|Core|Thread|NUMA|CPU\(s\)" >> gpurun_out/r6c_host.txt; cat gpurun_out/r6c_host.txt
timeout 900 python tools/dbg/c4_async.py -v > gpurun_out/r6c_c4_async.txt 2>&1; tail -90 gpurun_out/r6c_c4_async.txt
timeout 900 python -m pytest tests/test_gpu_pose_graph.py tests/test_gpu_fuzz_pose_graph.py -m gpu -q > gpurun_out/r6c_pg_tests.log 2>&1; tail -8 gpurun_out/r6c_pg_tests.log
OMP_PROC_BIND=close OMP_PLACES=cores timeout 600 python tools/cpu_threads.py 16 32 64 > gpurun_out/r6c_cpu_threads_bound.txt 2>&1; tail -8 gpurun_out/r6c_cpu_threads_bound.txt
OMP_PROC_BIND=spread OMP_PLACES=cores timeout 600 python tools/cpu_threads.py 16 32 64 > gpurun_out/r6c_cpu_threads_spread.txt 2>&1; tail -8 gpurun_out/r6c_cpu_threads_spread.txt
