export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rejected" 2>&1 | tail -15
STBA_LM_SPECULATE=0 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rejected" 2>&1 | grep -E "passed|failed"
