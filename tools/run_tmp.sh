export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed"
timeout 300 python tools/mega_stress.py 1500 20 2 2>&1 | tail -1
for f in 0 1 0 1; do
STBA_BWD_FUSED=$f timeout 300 python bench.py --reps 3 --steps 50 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_d3.json; python - $f <<'PY'
import json,sys
d=json.loads(open('gpurun_out/bench_d3.json').read())
print('FUSED', sys.argv[1], 'it/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],3), 'bwd', round(d['cholesky_ms']['backward'],4), 'factor', round(d['cholesky_ms']['factor_persistent_kernel'],3))
PY
done
