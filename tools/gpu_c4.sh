# usage: bash tools/gpu_c4.sh <tag>   -- C4 (pose graph) evidence: bench line + rocprofv3 kernel summary of the same command
export TMPDIR=/tmp
TAG=${1:-tmp}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
python $R/bench.py --config c4 > $O/${TAG}_c4_bench.json 2> $O/${TAG}_c4_bench.err
P4=/tmp/prof4_$TAG; rm -rf $P4; mkdir -p $P4
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P4 -- python $R/bench.py --config c4 --reps 1 --no-cpu-baseline > $P4/bench.json 2> $P4/bench.err)
find $P4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_c4_kernel_stats.csv
head -30 $O/${TAG}_c4_kernel_stats.csv | cut -c1-150
tail -3 $O/${TAG}_c4_bench.err
python - <<PY
import json
d=json.loads(open("$O/${TAG}_c4_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "lm_iterations", "pcg_iterations", "pcg_hit_cap", "final_cost")})
print("exact", d["exact_steps"]); print("gate", d.get("matched_result_gate")); print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("speedup_vs_cpu_port"))
PY
