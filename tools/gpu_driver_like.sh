# what the driver runs at round end, in one call: the GPU tests, smoke(), the default bench line (wall time of each)
export TMPDIR=/tmp
mkdir -p gpurun_out
S=$(date +%s); timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r6_final_pytest_gpu.txt 2>&1; tail -2 gpurun_out/r6_final_pytest_gpu.txt; echo "pytest wall $(( $(date +%s) - S )) s"
S=$(date +%s); python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -2; echo "smoke wall $(( $(date +%s) - S )) s"
S=$(date +%s); python bench.py > gpurun_out/r6_final_bench.json 2> gpurun_out/r6_final_bench.err; echo "bench wall $(( $(date +%s) - S )) s"; python -c "
import json; d=json.loads(open('gpurun_out/r6_final_bench.json').read().strip().splitlines()[-1]); print({k: d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','dtype','scaling','vs_baseline')}); print(d['roofline']['frac'], d['cpu_baseline']['value'], d['drop_in']['seconds']['solve'], {k: v['ms_median'] for k, v in d['published_workload'].items() if isinstance(v, dict) and v.get('ms_median')})"
