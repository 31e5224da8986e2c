"""Landmark-heavy scene (100 cameras x 1 000 000 landmarks) and one eighth of it: the Schur step with the rows cut by landmark range
(round 6): time per launch, pair rate, run-to-run bits, LM iterations / s; the reduced system of a smaller few-camera scene against the oracle."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
st = importlib.import_module("slam-tricks_amd")
scenes = importlib.import_module("slam-tricks_amd.scenes")
sharding = importlib.import_module("slam-tricks_amd.sharding")
import oracle_py as O
s = scenes.st20_scene(n_cams=40, n_pts=100000, max_obs_per_pt=10, seed=5, pix_noise=1e-3, retriangulate=False)
e = st.BAEngine(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
o = O.BA(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
e.evaluate(); e.normal_blocks()
_, ro, Jco, Jpo = o.evaluate()
rng = np.random.default_rng(8)
dc = rng.uniform(0.01, 0.1, (e.nc, 6)); dp = rng.uniform(0.01, 0.1, (e.np_, 3))
S1, r1 = e.reduced_system(dc, dp); S2, r2 = e.reduced_system(dc, dp)
So, rhso = o.reduced_system(ro, Jco, Jpo, dc, dp)
print("40 x 100000: S vs oracle %.2e rhs %.2e repeatable %s" % (np.abs(np.tril(S1) - np.tril(So)).max() / np.abs(So).max(),
      np.abs(r1 - rhso).max() / np.abs(rhso).max(), np.array_equal(S1, S2) and np.array_equal(r1, r2)))
Hcc, gc, _, _ = e.normal_blocks()
ms, at, pr = e.time_schur(reps=20)
print("40 x 100000: schur %.3f ms, %.2f G pairs/s" % (ms, pr / ms / 1e6))
s = scenes.st20_scene(n_cams=100, n_pts=1000000, max_obs_per_pt=10, seed=20, pix_noise=1e-3, retriangulate=False)
for name, sc in (("100 x 1e6", s), ("shard 0 of 8", sharding.make_shard(s, 0, 8))):
    e = st.BAEngine(sc["cams0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_feat"], sc["cam_fixed"])
    ms, at, pr = e.time_schur(reps=20)
    print("%s: schur %.3f ms, %.3g pairs -> %.2f G pairs/s" % (name, ms, pr, pr / ms / 1e6))
    e.set_params(sc["cams0"], sc["pts0"]); e.lm_iterations(3); e.set_params(sc["cams0"], sc["pts0"])
    t0 = time.perf_counter(); e.lm_iterations(20); dt = time.perf_counter() - t0
    print("%s: %.3f ms per LM iteration = %.1f LM it/s" % (name, 1e3 * dt / 20, 20 / dt))
    if name.startswith("100"):
        e.set_params(sc["cams0"], sc["pts0"]); a, ta = e.lm_iterations(5); ca = e.get_params()[0]
        e.set_params(sc["cams0"], sc["pts0"]); b, tb = e.lm_iterations(5); cb = e.get_params()[0]
        print("two runs of 5 iterations: same bits", np.array_equal(ta, tb) and np.array_equal(ca, cb))
