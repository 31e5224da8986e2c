export TMPDIR=/tmp
for rep in 1 2 3 4; do timeout 300 python tools/mega_trace.py run 6000; done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_sharding.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
timeout 300 python tools/mega_stress.py 3000 10 1 2>&1 | tail -1
timeout 300 python tools/mega_stress.py 1500 20 2 2>&1 | tail -1
STBA_MEGA_TRACE=/tmp/mega.bin timeout 300 python tools/mega_trace.py run 6000
timeout 100 python tools/mega_trace.py /tmp/mega.bin > gpurun_out/mega_trace_d3.txt
timeout 300 python bench.py --reps 3 --steps 50 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_d3.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_d3.json').read())
print('it/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],3), d['reps_ms_per_step'])
print(d['cholesky_ms']['factor_persistent_kernel'])
PY
