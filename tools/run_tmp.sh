export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_tmp.txt 2>&1
grep -E "passed|failed" gpurun_out/pytest_tmp.txt
python __graft_entry__.py smoke 2>&1 | tail -1
timeout 300 python tools/mega_stress.py 6000 6 1 2>&1 | tail -1
timeout 300 python tools/mega_stress.py 1500 20 2 2>&1 | tail -1
bash tools/gpu_prof.sh r2_f > gpurun_out/prof_r2_f.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/bench_r2_f_full.json 2> gpurun_out/bench_r2_f_full.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r2_f_full.json').read().strip().splitlines()[-1])
print('it/s', d['value'], 'ms', d['ms_per_step'], d['reps_ms_per_step'])
print(d['phase_ms_per_step']); print(d['cholesky_ms']['factor_persistent_kernel'], d['cholesky_ms']['backward']); print(d['roofline']['frac'], d['roofline']['traffic'], d['roofline_jacobian']['frac']); print(d['cpu_baseline']['value'], d['speedup_vs_cpu_port'], d['matched_result_gate']['passed'])
PY
