"""C4 under other forcing sequences (forcing_eta0, forcing_eta_min): LM it/s, PCG iterations, distance of the converged poses from the
exact-step run (gate 1e-5)."""
import importlib, json, os, sys, time
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
d = e.pcg_options()
print("defaults: eta0 %g eta_min %g" % (d.forcing_eta0, d.forcing_eta_min))
for kw in (dict(), dict(forcing_eta_min=3e-3), dict(forcing_eta_min=1e-2), dict(forcing_eta_min=3e-2), dict(forcing_eta_min=1e-1),
           dict(forcing_eta0=0.3), dict(forcing_eta0=0.5), dict(forcing_eta0=0.3, forcing_eta_min=1e-2), dict(forcing_eta0=0.03), dict(forcing_eta0=0.01)):
    times = []
    for rep in range(4):
        e = fresh(); e.solve(max_num_iterations=1, pcg=e.pcg_options(**kw))
        e = fresh(); t0 = time.perf_counter(); summ, tr, tot = e.solve(pcg=e.pcg_options(**kw)); times.append(time.perf_counter() - t0)
    p = e.get_poses()
    print("%-55s it %2d pcg %4d %.3f ms %.0f LM it/s cost rel %.1e dist exact %.2e" % (kw, summ.num_iterations, tot, 1e3 * np.median(times),
          summ.num_iterations / np.median(times), abs(summ.final_cost - sx.final_cost) / sx.final_cost, pdiff(p, px)))
