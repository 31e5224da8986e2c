export TMPDIR=/tmp
STBA_DEBUG_KNOBS=1 python -c "
import importlib; b=importlib.import_module('slam-tricks_amd.build'); print(b.build(force=True))" > /tmp/build.log 2>&1; tail -1 /tmp/build.log
STBA_CREATE_TIMING=1 python - <<'PY' 2>&1 | grep -v amdgpu
import importlib, time, numpy as np, os
st = importlib.import_module("slam-tricks_amd"); scenes = importlib.import_module("slam-tricks_amd.scenes")
s = scenes.st20_scene(n_cams=1000, n_pts=100000, max_obs_per_pt=10, seed=20, pix_noise=1e-3)
for k in range(3):
    t0 = time.perf_counter()
    e = st.BAEngine(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
    print("create total %.1f ms" % (1e3 * (time.perf_counter() - t0)))
    del e
PY
