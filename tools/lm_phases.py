"""per-phase device times of the C5 LM iteration (bench.py's workload), one line; usage: python tools/lm_phases.py [steps]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
st = importlib.import_module("slam-tricks_amd")
class A: cams = 1000; pts = 100000; obs_per_pt = 10
s = bench.load_scene(A, 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
eng = st.BAEngine(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
eng.lm_iterations(3)
import time
best = None
for rep in range(3):
    eng.set_params(s["cams0"], s["pts0"])
    t0 = time.perf_counter()
    summ, tr = eng.lm_iterations(steps)
    dt = (time.perf_counter() - t0) / steps * 1e3
    eng.set_params(s["cams0"], s["pts0"])
    summ, tr = eng.lm_iterations(steps, phase_timing=1)
    ph = {k: getattr(summ, k) / steps for k in ("ms_linearize", "ms_schur", "ms_solve", "ms_backsub", "ms_cost")}
    if best is None or dt < best[0]: best = (dt, ph, summ.final_cost)
dt, ph, fc = best
print(f"ms/step {dt:.4f}  " + "  ".join(f"{k[3:]} {v:.4f}" for k, v in ph.items()) + f"  non-chol {ph['ms_linearize'] + ph['ms_schur'] + ph['ms_backsub'] + ph['ms_cost']:.4f}  final cost {fc:.12e}")
