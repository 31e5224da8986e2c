import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
st = importlib.import_module("slam-tricks_amd")
scenes = importlib.import_module("slam-tricks_amd.scenes")
s = scenes.st20_scene(pix_noise=1e-3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
bad = 0
ref = None
for rep in range(n):
    kw = {} if rep % 2 else dict(function_tolerance_takes_step=0)
    e = st.BAEngine(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
    summ, tr = e.solve(**kw)
    key = (rep % 2, summ.num_iterations, summ.termination_reason, tr[:, 0].tobytes())
    if ref is None: ref = {}
    if rep % 2 not in ref: ref[rep % 2] = key
    if key != ref[rep % 2]:
        bad += 1
        print("ANOMALY rep", rep, kw, summ.as_dict(), tr[:, [0, 2, 3, 6]])
print("runs", n, "anomalies", bad)
