"""runs a few fixed-work LM iterations of the C5 problem and nothing else (for PMC counter passes on the assembly kernels)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
st = importlib.import_module("slam-tricks_amd")
scenes = importlib.import_module("slam-tricks_amd.scenes")
cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), "stba_scene_c1000_p100000_m10_s20.npz")
if os.path.exists(cache):
    z = np.load(cache); s = {k: z[k] for k in z.files}
else:
    s = scenes.st20_scene(n_cams=1000, n_pts=100000, max_obs_per_pt=10, seed=20, pix_noise=1e-3)
    np.savez(cache, **s)
e = st.BAEngine(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
summ, tr = e.lm_iterations(int(sys.argv[1]) if len(sys.argv) > 1 else 3, phase_timing=1)
print("ms", {k: getattr(summ, k) for k in ("ms_linearize", "ms_schur", "ms_solve", "ms_backsub", "ms_cost")})
