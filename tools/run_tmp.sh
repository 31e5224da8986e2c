export TMPDIR=/tmp
timeout 900 python bench.py --reps 3 --steps 50 > gpurun_out/bench_lapack.json 2> gpurun_out/bench_lapack.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_lapack.json').read().strip().splitlines()[-1])
print('it/s', d['value'])
for k in ('cpu_baseline','cpu_baseline_plain_c_cholesky','cpu_baseline_lapack_cholesky','cpu_baseline_single_thread'):
    if k in d: print(k, d[k]['value'], d[k]['cores'], d[k]['sample'])
print(d.get('speedup_vs_cpu_port'), d['matched_result_gate']['passed'])
PY
tail -3 gpurun_out/bench_lapack.err
