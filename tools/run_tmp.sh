export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "calib" > gpurun_out/pytest_tmp.txt 2>&1
tail -15 gpurun_out/pytest_tmp.txt
python - <<'PY'
import importlib, time, sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
st = importlib.import_module("slam-tricks_amd"); scenes = importlib.import_module("slam-tricks_amd.scenes")
s = scenes.calib_scene(n_views=20, rows=8, cols=11, seed=3)
p0 = np.concatenate([s["intr_true"][:4] * (1 + 5e-3), np.zeros(5), s["xis_true"].reshape(-1)])
st.calib_gauss_newton(p0, s["obj"], s["img"], 10)
t = time.time()
for _ in range(20): p, it, tr = st.calib_gauss_newton(p0, s["obj"], s["img"], 10)
print("C3 GN, 10 iterations max: %.3f ms per solve (incl. create/upload/readback), iterations %d" % ((time.time() - t) / 20 * 1e3, it), tr)
PY
