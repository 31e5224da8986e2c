export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_cpp_shim.py -m gpu -x -q 2>&1 | tail -40
