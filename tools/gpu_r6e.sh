# round 6, fifth GPU call: C4 with the coarse matrix built on the second stream; coarse-residual stopping test
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python tools/dbg/c4_async.py > gpurun_out/r6e_c4_async.txt 2>&1; tail -16 gpurun_out/r6e_c4_async.txt
