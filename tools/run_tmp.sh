export TMPDIR=/tmp
for rep in 1 2 3 4; do timeout 300 python tools/mega_trace.py run 6000; done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
bash tools/pmc_chol.sh r2_nt > gpurun_out/pmc_chol_r2_nt.log 2>&1
tail -1 gpurun_out/pmc_chol_r2_nt.log | cut -c1-200
