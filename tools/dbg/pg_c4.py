import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
st = importlib.import_module("slam-tricks_amd")
scenes = importlib.import_module("slam-tricks_amd.scenes")
s = scenes.pose_graph_scene(n_nodes=10000, loops_per_node=3, seed=4)
KWS = (dict(), dict(one_kernel_solve=0), dict(forcing_eta0=0.0), dict(forcing_eta0=0.0, one_kernel_solve=0)) if os.environ.get('PG_AB') else (dict(), dict(coarse_refresh_every=2), dict(coarse_refresh_every=3), dict(coarse_refresh_every=4), dict(coarse_refresh_every=100), dict(check_every=2), dict(check_every=6))
for kw in KWS:
    best = None
    for rep in range(3):
        e = st.PGEngine(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
        opt = st.default_options(minimizer_progress_to_stdout=int(os.environ.get("PG_PROGRESS", "0")) if rep == 0 else 0)
        t0 = time.time()
        summ, tr, tot = e.solve(opt=opt, pcg=e.pcg_options(**kw))
        dt = time.time() - t0
        if best is None or dt < best[0]: best = (dt, summ.num_iterations, tot, summ.final_cost, e.pcg_summary().as_dict())
    print(kw, "sec %.4f LM it %d (%.1f it/s) pcg %d final %.9f" % (best[0], best[1], best[1] / best[0], best[2], best[3]), best[4], flush=True)
