export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_tmp.txt 2>&1
grep -E "passed|failed" gpurun_out/pytest_tmp.txt
timeout 300 python tools/mega_stress.py 3000 10 1 2>&1 | tail -1
timeout 300 python tools/mega_stress.py 1500 20 2 2>&1 | tail -1
timeout 300 python bench.py --reps 3 --steps 50 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_d3.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_d3.json').read())
print('it/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],3), d['reps_ms_per_step'])
print(d['cholesky_ms']['factor_persistent_kernel'], d['roofline']['frac'])
PY
