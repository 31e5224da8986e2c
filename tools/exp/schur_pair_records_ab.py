"""Round 6: the pair form of the Schur complement from per-observation records (STBA_SCHUR_PAIRS_RECORDS) against the product form
(STBA_SCHUR_PAIRS): time per Schur step, the reduced system against the other form, run-to-run bits.  C5 and the landmark-heavy scene."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
st = importlib.import_module("slam-tricks_amd")
scenes = importlib.import_module("slam-tricks_amd.scenes")
which = sys.argv[1:] or ["small", "c5", "lh"]
for w in which:
    if w == "small": s = scenes.st20_scene()
    elif w == "c5": s = scenes.st20_scene(n_cams=1000, n_pts=100000, max_obs_per_pt=10, seed=20, pix_noise=1e-3)
    else: s = scenes.st20_scene(n_cams=100, n_pts=1000000, max_obs_per_pt=10, seed=20, pix_noise=1e-3, retriangulate=False)
    e = st.BAEngine(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
    res = {}
    for mode in (1, 3):
        e.set_schur_mode(mode)
        ms, at, pr = e.time_schur(reps=20)
        e.set_params(s["cams0"], s["pts0"])
        e.evaluate(residuals=False, jac=False); e.normal_blocks()
        dc = np.full(6 * len(s["cams0"]), 1e-4); dp = np.full(3 * len(s["pts0"]), 1e-4)
        if w == "lh" or True:
            S1, r1 = e.reduced_system(dc, dp)
            S2, r2 = e.reduced_system(dc, dp)
        res[mode] = (ms, S1, r1, np.array_equal(S1, S2) and np.array_equal(r1, r2))
        print(w, "mode", mode, "schur %.4f ms  pairs %.3g  -> %.2f G pairs/s  repeatable %s" % (ms, pr, pr / ms / 1e6, res[mode][3]))
    dS = np.abs(np.tril(res[1][1]) - np.tril(res[3][1])).max() / np.abs(res[1][1]).max()
    dr = np.abs(res[1][2] - res[3][2]).max() / np.abs(res[1][2]).max()
    print(w, "forms differ: S %.2e rhs %.2e (relative to the largest entry)" % (dS, dr))
    if w != "lh":
        for mode in (1, 3):
            e.set_params(s["cams0"], s["pts0"]); e.set_schur_mode(mode)
            t0 = time.perf_counter(); summ, tr = e.solve(); dt = time.perf_counter() - t0
            print(w, "mode", mode, "solve: it %d cost %.12g  %.1f ms" % (summ.num_iterations, summ.final_cost, 1e3 * dt))
    for mode in (1, 3):
        e.set_params(s["cams0"], s["pts0"]); e.set_schur_mode(mode)
        e.lm_iterations(3)
        e.set_params(s["cams0"], s["pts0"])
        t0 = time.perf_counter(); e.lm_iterations(20); dt = time.perf_counter() - t0
        print(w, "mode", mode, "20 LM iterations: %.3f ms each = %.1f LM it/s" % (1e3 * dt / 20, 20 / dt))
