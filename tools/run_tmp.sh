export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
python - <<'PY'
import importlib
st = importlib.import_module("slam-tricks_amd")
print("wide on ", st.cholesky_time_split(6000, reps=8))
PY
STBA_BWD_WIDE=0 python - <<'PY'
import importlib
st = importlib.import_module("slam-tricks_amd")
print("wide off", st.cholesky_time_split(6000, reps=8))
PY
