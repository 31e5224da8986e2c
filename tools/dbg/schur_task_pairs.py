import importlib, os, sys, numpy as np
sys.path.insert(0, os.getcwd())
st = importlib.import_module("slam-tricks_amd")
scenes = importlib.import_module("slam-tricks_amd.scenes")
cache = "/tmp/c5scene.npz"
if os.path.exists(cache):
    s = dict(np.load(cache))
else:
    s = scenes.st20_scene(n_cams=1000, n_pts=100000, max_obs_per_pt=10, seed=20, pix_noise=1e-3); np.savez(cache, **s)
e = st.BAEngine(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
e.evaluate(); e.normal_blocks()
for r in range(3):
    ms, at, pr = e.time_schur(reps=30)
print("STBA_SCHUR_TASK_PAIRS", os.environ.get("STBA_SCHUR_TASK_PAIRS"), "schur %.4f ms" % ms, flush=True)
