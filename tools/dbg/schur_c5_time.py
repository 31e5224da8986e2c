"""C5 (1000 cameras x 100 000 landmarks): the Schur kernel alone (hipEvents, per launch), run-to-run bits of S, S against the oracle on a
small scene, LM iterations / s.  Usage: python tools/dbg/schur_c5_time.py"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
st = importlib.import_module("slam-tricks_amd")
scenes = importlib.import_module("slam-tricks_amd.scenes")
import oracle_py as O
s = scenes.st20_scene(n_cams=60, n_pts=4000, max_obs_per_pt=12, seed=5, pix_noise=1e-3)
e = st.BAEngine(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
o = O.BA(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
e.evaluate(); e.normal_blocks()
_, ro, Jco, Jpo = o.evaluate()
rng = np.random.default_rng(8)
dc = rng.uniform(0.01, 0.1, (e.nc, 6)); dp = rng.uniform(0.01, 0.1, (e.np_, 3))
S1, r1 = e.reduced_system(dc, dp); S2, r2 = e.reduced_system(dc, dp)
So, rhso = o.reduced_system(ro, Jco, Jpo, dc, dp)
print("60 x 4000: S vs oracle %.2e rhs %.2e repeatable %s" % (np.abs(np.tril(S1) - np.tril(So)).max() / np.abs(So).max(),
      np.abs(r1 - rhso).max() / np.abs(rhso).max(), np.array_equal(S1, S2) and np.array_equal(r1, r2)))
s = scenes.st20_scene(n_cams=1000, n_pts=100000, max_obs_per_pt=10, seed=20, pix_noise=1e-3)
e = st.BAEngine(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
for rep in range(3):
    ms, at, pr = e.time_schur(reps=30)
    print("C5: schur %.4f ms, %.3g pairs -> %.2f G pairs/s, %.3g LDS atomics" % (ms, pr, pr / ms / 1e6, at))
e.set_params(s["cams0"], s["pts0"]); e.lm_iterations(3); e.set_params(s["cams0"], s["pts0"])
t0 = time.perf_counter(); e.lm_iterations(50); dt = time.perf_counter() - t0
print("C5: %.3f ms per LM iteration = %.1f LM it/s" % (1e3 * dt / 50, 50 / dt))
