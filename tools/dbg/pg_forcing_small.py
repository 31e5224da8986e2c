"""the forcing floor on other graphs than C4: distance of the converged poses from the oracle's exact-step answer (gate 1e-5)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
st = importlib.import_module("slam-tricks_amd")
scenes = importlib.import_module("slam-tricks_amd.scenes")
def pdiff(a, b):
    dq = np.minimum(np.abs(a[:, :4] - b[:, :4]).max(1), np.abs(a[:, :4] + b[:, :4]).max(1)).max()
    return float(max(dq, np.abs(a[:, 4:] - b[:, 4:]).max()))
for name, kw in (("400 nodes, 2x noise", dict(n_nodes=400, loops_per_node=3, seed=4, sigma_t=0.02, sigma_r=0.004, turns=6)),
                 ("3000 nodes", dict(n_nodes=3000, loops_per_node=3, seed=5, turns=8)),
                 ("2000 nodes, 3x noise", dict(n_nodes=2000, loops_per_node=2, seed=9, sigma_t=0.03, sigma_r=0.006, turns=5)),
                 ("10000 nodes, 1 closure", dict(n_nodes=10000, loops_per_node=1, seed=6))):
    s = scenes.pose_graph_scene(**kw)
    e = st.PGEngine(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
    sx, _, nx = e.solve(pcg=e.pcg_options(forcing_eta0=0.0, relative_tolerance=1e-12, max_iterations=4000))
    px = e.get_poses()
    row = []
    for fl in (1e-10, 1e-3, 3e-3, 1e-2, 3e-2, 1e-1):
        e = st.PGEngine(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
        summ, tr, tot = e.solve(pcg=e.pcg_options(forcing_eta_min=fl))
        row.append("%g: it %d pcg %d %.1e" % (fl, summ.num_iterations, tot, pdiff(e.get_poses(), px)))
    print(name, "(exact: it %d pcg %d) | " % (sx.num_iterations, nx) + " | ".join(row), flush=True)
