export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r6h_pytest.log 2>&1; tail -4 gpurun_out/r6h_pytest.log
for k in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -1; done > gpurun_out/r6h_parity_x3.log; cat gpurun_out/r6h_parity_x3.log
timeout 900 python bench.py --second-scene --no-cpu-baseline --no-library-baseline --no-drop-in > gpurun_out/r6h_second_scene.json 2> gpurun_out/r6h_second_scene.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6h_second_scene.json').read().strip().splitlines()[-1])
lh=d['landmark_heavy']; print('C5', d['value'], 'lh', lh['value'], lh.get('ms_per_step'), lh.get('phase_ms_per_step'), lh.get('predicted_scaling',{}).get('speedup_at_8_best'))
PY
