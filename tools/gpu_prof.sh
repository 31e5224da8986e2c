# usage: bash tools/gpu_prof.sh <tag> ; writes gpurun_out/prof_<tag>/ (kernel trace + stats, csv)
export TMPDIR=/tmp
TAG=${1:-tmp}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -40 {}'
# keep only the small summaries (the full trace is large)
find $OUT -name "*kernel_trace.csv" -size +20M -delete
