export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_tmp.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_tmp.log; grep -E "^(FAILED|ERROR)|Error|assert " gpurun_out/pytest_tmp.log | head -12
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_tmp.json 2> gpurun_out/bench_tmp.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_tmp.json').read().strip().splitlines()[-1])
print('it/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],3))
print('phase', {k:round(v,3) for k,v in d['phase_ms_per_step'].items() if isinstance(v, float)})
c=d['cholesky_ms']
print('chol factor', round(c['factor_persistent_kernel'],3), 'bwd', round(c['backward'],3), 'stages', {k:round(v,3) for k,v in c['stage_kernels_serial'].items()})
print('jac GB/s', round(d['roofline_jacobian']['achieved'],1), 'chol TF', round(d['roofline_mfma']['achieved'],2))
PY
tail -3 gpurun_out/bench_tmp.err
