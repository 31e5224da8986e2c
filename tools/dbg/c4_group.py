"""C4 under other sizes of the coarse space (coarse_group = nodes per group)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
st = importlib.import_module("slam-tricks_amd")
scenes = importlib.import_module("slam-tricks_amd.scenes")
s = scenes.pose_graph_scene(n_nodes=10000, loops_per_node=3, seed=4)
def pdiff(a, b):
    dq = np.minimum(np.abs(a[:, :4] - b[:, :4]).max(1), np.abs(a[:, :4] + b[:, :4]).max(1)).max()
    return float(max(dq, np.abs(a[:, 4:] - b[:, 4:]).max()))
def fresh():
    return st.PGEngine(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
e = fresh()
sx, trx, nx = e.solve(pcg=e.pcg_options(forcing_eta0=0.0, relative_tolerance=1e-12, max_iterations=2000))
px = e.get_poses()
for kw in (dict(), dict(coarse_group=128), dict(coarse_group=256), dict(coarse_group=64, coarse_async=0), dict(coarse_group=128, coarse_async=0)):
    times = []
    for rep in range(4):
        e = fresh(); e.solve(max_num_iterations=1, pcg=e.pcg_options(**kw))
        e = fresh(); t0 = time.perf_counter(); summ, tr, tot = e.solve(pcg=e.pcg_options(**kw)); times.append(time.perf_counter() - t0)
    p = e.get_poses(); ps = e.pcg_summary().as_dict()
    print("%-45s it %2d pcg %4d %.3f ms %.0f LM it/s dist exact %.2e coarse dim %d one-kernel %d" % (kw, summ.num_iterations, tot, 1e3 * np.median(times),
          summ.num_iterations / np.median(times), pdiff(p, px), ps["coarse_dim"], ps["one_kernel_solves"]))
