export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 900 python bench.py > gpurun_out/bench_r2a.json 2> gpurun_out/bench_r2a.err; tail -3 gpurun_out/bench_r2a.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r2a.json').read().strip().splitlines()[-1])
print('it/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],3), d['reps_ms_per_step'])
print('phase', {k:round(v,3) for k,v in d['phase_ms_per_step'].items()})
print('gate', d.get('matched_result_gate'))
print('cpu', d.get('cpu_baseline'))
print('ceres', d.get('ceres_baseline'))
PY
