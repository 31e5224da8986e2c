# scratch script for one-off gpurun calls (gpurun ships the repository, not gpurun_out/): edit, run with
#   gpurun --timeout N -- 'bash tools/run_tmp.sh'
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed"
