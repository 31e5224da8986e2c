# usage: bash tools/gpu_round.sh <tag>   -- the evidence of a round in one call: full bench line, rocprofv3 kernel summary of the
# same command, kernel timeline of one LM iteration, C4 line + its kernel summary.  Everything lands in gpurun_out/<tag>_*.
export TMPDIR=/tmp
TAG=${1:-tmp}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
python $R/bench.py > $O/${TAG}_bench_full.json 2> $O/${TAG}_bench_full.err
P=/tmp/prof_$TAG; rm -rf $P; mkdir -p $P
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P -- python $R/bench.py --steps 20 --reps 2 --warmup 2 --no-cpu-baseline > $P/bench.json 2> $P/bench.err)
find $P -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_kernel_stats.csv
bash $R/tools/gpu_iter_trace.sh > $O/${TAG}_iter_trace.txt 2>&1
python $R/bench.py --config c4 > $O/${TAG}_c4_bench.json 2> $O/${TAG}_c4_bench.err
P4=/tmp/prof4_$TAG; rm -rf $P4; mkdir -p $P4
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P4 -- python $R/bench.py --config c4 --reps 1 --no-cpu-baseline > $P4/bench.json 2> $P4/bench.err)
find $P4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_c4_kernel_stats.csv
head -12 $O/${TAG}_kernel_stats.csv | cut -c1-160
tail -3 $O/${TAG}_bench_full.err
python - <<PY
import json
d=json.loads(open("$O/${TAG}_bench_full.json").read().strip().splitlines()[-1])
print('it/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],4), 'phases', {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['phase_ms_per_step'].items()})
print('roofline frac', d['roofline']['frac'], 'jac frac', d['roofline_jacobian']['frac'], 'cpu', d['cpu_baseline']['value'], 'speedup', d.get('speedup_vs_cpu_port'), 'gate', d['matched_result_gate']['passed'])
PY
