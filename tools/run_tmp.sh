export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_tmp.txt 2>&1
grep -E "passed|failed" gpurun_out/pytest_tmp.txt
python __graft_entry__.py smoke 2>&1 | tail -1
bash tools/gpu_prof.sh r2_d > gpurun_out/prof_r2_d.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/bench_r2_d_full.json 2> gpurun_out/bench_r2_d_full.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r2_d_full.json').read().strip().splitlines()[-1])
print('it/s', d['value'], 'ms', d['ms_per_step'], d['reps_ms_per_step'])
print(d['phase_ms_per_step']); print(d['cholesky_ms']); print(d['roofline']['frac'], d['roofline_jacobian']['frac']); print(d['cpu_baseline']['value'], d['speedup_vs_cpu_port'], d['matched_result_gate']['passed'])
PY
