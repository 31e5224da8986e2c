export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_tmp.txt 2>&1; grep -n "passed\|failed" gpurun_out/pytest_tmp.txt
timeout 900 python bench.py > gpurun_out/bench_r2b.json 2> gpurun_out/bench_r2b.err; tail -2 gpurun_out/bench_r2b.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r2b.json').read().strip().splitlines()[-1])
print('it/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],3))
print('phase', {k:round(v,3) for k,v in d['phase_ms_per_step'].items()})
print('jac', d['roofline_jacobian']['achieved'], d['roofline_jacobian']['ms_per_launch'], 'chol', d['cholesky_ms']['factor_persistent_kernel'], d['cholesky_ms']['backward'])
print('gate', d.get('matched_result_gate',{}).get('passed'))
PY
bash tools/gpu_prof.sh r2_b > gpurun_out/prof_r2_b.log 2>&1; head -30 gpurun_out/prof_r2_b.log
