export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_tmp.txt 2>&1
tail -3 gpurun_out/pytest_tmp.txt
