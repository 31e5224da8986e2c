export TMPDIR=/tmp
bash tools/gpu_prof.sh r2_c > gpurun_out/prof_r2_c.log 2>&1
bash tools/pmc_jacobian.sh r2_c > gpurun_out/pmc_jac_r2_c.log 2>&1
bash tools/pmc_chol.sh r2_c > gpurun_out/pmc_chol_r2_c.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/bench_r2_c_full.json 2> gpurun_out/bench_r2_c_full.err
tail -c 600 gpurun_out/bench_r2_c_full.json
