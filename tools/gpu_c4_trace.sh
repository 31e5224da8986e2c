export TMPDIR=/tmp
OUT=/tmp/c4_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/tools/c4_iter_trace.py run > $OUT/run.log 2>&1
tail -1 $OUT/run.log
python $GRAFT_REPO_ROOT/tools/c4_iter_trace.py $OUT
