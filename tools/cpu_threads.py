"""times the oracle's fixed-work LM iteration on the C5 problem for several OpenMP thread counts"""
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
for nt in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64]:
    ba = O.BA(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
    t = time.time(); summ, _ = ba.solve(fixed_iterations=1, num_threads=nt); dt = time.time() - t
    print(nt, round(dt, 2), "lin", round(summ.seconds_linearize, 2), "schur", round(summ.seconds_schur, 2),
          "solve", round(summ.seconds_solve, 2), flush=True)
