"""device time of the Schur-complement kernel on bench.py's C5 scene, one line; usage: python tools/schur_time.py [reps]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
st = importlib.import_module("slam-tricks_amd")
class A: cams = 1000; pts = 100000; obs_per_pt = 10
s = bench.load_scene(A, 0)
eng = st.BAEngine(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
summ, tr = eng.lm_iterations(3)
ms, atomics, pairs = eng.time_schur(int(sys.argv[1]) if len(sys.argv) > 1 else 20)
print(f"schur {ms:.4f} ms  atomics {atomics:.4g}  pairs {pairs:.4g}  cost after 3 its {summ.final_cost:.12e}")
