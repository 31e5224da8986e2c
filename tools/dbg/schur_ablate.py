import importlib, os, sys
sys.path.insert(0, os.getcwd())
st = importlib.import_module("slam-tricks_amd")
scenes = importlib.import_module("slam-tricks_amd.scenes")
s = scenes.st20_scene(n_cams=1000, n_pts=100000, max_obs_per_pt=10, seed=20, pix_noise=1e-3)
e = st.BAEngine(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
e.evaluate(); e.normal_blocks()
for ab in (0, 1, 2, 3, 0):
    os.environ["STBA_SCHUR_ABLATE"] = str(ab)
    for r in range(2):
        ms, at, pr = e.time_schur(reps=30)
    print("ablate %d (1: gathers from 64 hot records, 2: no LDS atomics): schur %.4f ms" % (ab, ms), flush=True)
