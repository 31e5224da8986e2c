"""stba_ba_create at C5 phase by phase (debug build: STBA_CREATE_TIMING=1 prints the phases on stderr), three creations in one process."""
import importlib, os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
st = importlib.import_module("slam-tricks_amd")
scenes = importlib.import_module("slam-tricks_amd.scenes")
cache = "/tmp/c5scene.npz"
if os.path.exists(cache): s = dict(np.load(cache))
else:
    s = scenes.st20_scene(n_cams=1000, n_pts=100000, max_obs_per_pt=10, seed=20, pix_noise=1e-3); np.savez(cache, **s)
for rep in range(3):
    t0 = time.perf_counter()
    e = st.BAEngine(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
    print("create %d: %.1f ms" % (rep, 1e3 * (time.perf_counter() - t0)), file=sys.stderr, flush=True)
    del e
