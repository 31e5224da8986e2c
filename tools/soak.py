"""soak run: thousands of factorisations at several sizes and 1200 LM iterations of the C5 problem; prints the dependency time-out count (must stay 0)
usage: python tools/soak.py [multiplier]"""
import importlib, sys, time
sys.path.insert(0, ".")
import numpy as np
st = importlib.import_module("slam-tricks_amd")
MUL = int(sys.argv[1]) if len(sys.argv) > 1 else 1
t0 = time.time()
for n in (6000, 4100, 3000, 2000, 1000, 777, 6000):
    ms = st.cholesky_time(n, reps=300 * MUL)
    print(n, round(ms, 4), "timeouts", st.cholesky_timeout_count(), flush=True)
import bench
class A: cams = 1000; pts = 100000; obs_per_pt = 10
s = bench.load_scene(A, 0)
eng = st.BAEngine(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
costs = set()
for k in range(6 * MUL):
    eng.set_params(s["cams0"], s["pts0"])
    summ, tr = eng.lm_iterations(200)
    costs.add(round(summ.final_cost, 9))
print(f"final costs of {6 * MUL} x 200 iterations:", sorted(costs)[:3], "...", len(costs), "distinct,", "timeouts", st.cholesky_timeout_count(), "wall", round(time.time() - t0, 1))
