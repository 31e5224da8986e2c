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
# pose graph (round 5): the one-kernel PCG solve hundreds of times -- no solve may give up (a stamp that never comes) and every run must
# end with the same bits
scenes = importlib.import_module("slam-tricks_amd.scenes")
g = scenes.pose_graph_scene(n_nodes=10000, loops_per_node=3, seed=4)
ref, gave_up, differ, t1 = None, 0, 0, time.time()
for k in range(200 * MUL):
    e = st.PGEngine(g["poses0"], g["edge_i"], g["edge_j"], g["meas"], g["node_fixed"])
    summ, tr, tot = e.solve(pcg=e.pcg_options(forcing_eta0=0.0 if k % 4 == 0 else 0.1))
    ps = e.pcg_summary()
    gave_up += ps.solves - ps.one_kernel_solves
    key = (k % 4 == 0, tr.tobytes(), e.get_poses().tobytes())
    if ref is None: ref = {}
    if key[0] not in ref: ref[key[0]] = key[1:]
    elif ref[key[0]] != key[1:]: differ += 1
print(f"pose graph C4, {200 * MUL} solves ({50 * MUL} with exact steps): solves that gave up {gave_up}, runs that differ from the first {differ}, wall {round(time.time() - t1, 1)}")
