export TMPDIR=/tmp
for k in 1 2 3 4 5 6; do timeout 600 python -m pytest tests/test_sharding.py -x -q -m gpu 2>&1 | grep -E "passed|failed"; done
