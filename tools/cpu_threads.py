"""times the oracle's fixed-work LM iterations on the C5 problem for several OpenMP thread counts, phase by phase (plain C Cholesky
and LAPACK): python tools/cpu_threads.py [threads ...]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle_py as O
scenes = importlib.import_module("slam-tricks_amd.scenes")
cache = "/tmp/stba_scene_c1000_p100000_m10_s20.npz"
if os.path.exists(cache):
    z = np.load(cache); s = {k: z[k] for k in z.files}
else:
    s = scenes.st20_scene(n_cams=1000, n_pts=100000, max_obs_per_pt=10, seed=20, pix_noise=1e-3)
    np.savez(cache, **s)
print("cores", os.cpu_count(), flush=True)
for nt in [int(a) for a in sys.argv[1:]] or [1, 8, 16, 32, 64, 128]:
    if nt > (os.cpu_count() or 1): continue
    for lap in (False, True):
        if lap and not O.use_lapack(True, threads=min(nt, 64)): continue
        try:
            ba = O.BA(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
            ba.solve(fixed_iterations=1, num_threads=nt)
            ba = O.BA(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
            n = 1 if nt == 1 else 3
            t = time.time(); summ, _ = ba.solve(fixed_iterations=n, num_threads=nt); dt = (time.time() - t) / n
            print("threads %3d %-7s %.3f s/it = %.2f it/s | lin %.3f schur %.3f solve %.3f backsub %.3f cost %.3f" % (
                nt, "lapack" if lap else "plain-c", dt, 1.0 / dt, summ.seconds_linearize / (n + 1), summ.seconds_schur / n, summ.seconds_solve / n,
                summ.seconds_backsub / n, summ.seconds_cost / n), flush=True)
        finally:
            if lap: O.use_lapack(False)
