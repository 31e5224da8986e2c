export TMPDIR=/tmp
bash tools/pmc_chol.sh r2_e > gpurun_out/pmc_chol_r2_e.log 2>&1
tail -1 gpurun_out/pmc_chol_r2_e.log | cut -c1-300
bash tools/gpu_prof.sh r2_e > gpurun_out/prof_r2_e.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/bench_r2_e_full.json 2> gpurun_out/bench_r2_e_full.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r2_e_full.json').read().strip().splitlines()[-1])
print('it/s', d['value'], 'ms', d['ms_per_step'], d['reps_ms_per_step'])
print(d['phase_ms_per_step']); print(d['cholesky_ms']); print(d['roofline']['frac'], d['roofline_jacobian']['frac']); print(d['cpu_baseline']['value'], d['speedup_vs_cpu_port'], d['matched_result_gate']['passed'])
PY
