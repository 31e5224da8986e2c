# usage: bash tools/gpu_trace.sh [n] ; timeline of one factorisation -> gpurun_out/chol_trace.txt
export TMPDIR=/tmp
N=${1:-6000}
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_tmp
rm -rf $OUT; mkdir -p $OUT
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/tools/chol_trace.py run $N > $OUT/run.log 2>&1
python $GRAFT_REPO_ROOT/tools/chol_trace.py $OUT $N > $GRAFT_REPO_ROOT/gpurun_out/chol_trace.txt
tail -2 $OUT/run.log
rm -rf $OUT
