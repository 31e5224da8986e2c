import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
st = importlib.import_module("slam-tricks_amd")
scenes = importlib.import_module("slam-tricks_amd.scenes")
s = scenes.st20_scene(pix_noise=1e-3)
for rep in range(3):
    for kw in ({}, dict(function_tolerance_takes_step=0)):
        e = st.BAEngine(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
        summ, tr = e.solve(**kw)
        print(kw, summ.num_iterations, summ.termination_type, summ.termination_reason, ["%.6e" % v for v in tr[:, 0]], [int(v) for v in tr[:, 6]], ["%.2e" % v for v in tr[:, 3]])
