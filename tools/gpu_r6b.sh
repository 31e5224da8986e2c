# round 6, second GPU call: C4 with the coarse inverse on its second stream; the record form of the Schur pair kernel
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python tools/dbg/c4_async.py -v > gpurun_out/r6b_c4_async.txt 2>&1; tail -60 gpurun_out/r6b_c4_async.txt
timeout 900 python -m pytest tests/test_gpu_pose_graph.py tests/test_gpu_fuzz_pose_graph.py -m gpu -x -q > gpurun_out/r6b_pg_tests.log 2>&1; tail -8 gpurun_out/r6b_pg_tests.log
timeout 1200 python tools/dbg/schur_forms6.py small c5 lh > gpurun_out/r6b_schur_forms.txt 2>&1; tail -40 gpurun_out/r6b_schur_forms.txt
timeout 600 python tools/cpu_threads.py 16 32 64 128 > gpurun_out/r6b_cpu_threads.txt 2>&1; tail -12 gpurun_out/r6b_cpu_threads.txt
