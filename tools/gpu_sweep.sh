# usage: bash tools/gpu_sweep.sh "VAR=a VAR=b ..." ; runs the short bench once per environment setting
export TMPDIR=/tmp
for kv in "$@"; do
  env $kv timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/sweep.json 2> gpurun_out/sweep.err
  python - "$kv" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/sweep.json').read().strip().splitlines()[-1])
print(sys.argv[1], 'it/s', round(d['value'],2), 'solve', round(d['phase_ms_per_step']['ms_solve'],3))
PY
done
