"""one C4 solve with exact steps (1187 PCG iterations): the per-iteration time of the PCG path in use; with a debug build and
STBA_PP_TIMING=1 the one-kernel solve prints its phases"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
st = importlib.import_module("slam-tricks_amd")
scenes = importlib.import_module("slam-tricks_amd.scenes")
s = scenes.pose_graph_scene(n_nodes=10000, loops_per_node=3, seed=4)
for rep in range(3):
    e = st.PGEngine(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
    t0 = time.time()
    summ, tr, tot = e.solve(pcg=e.pcg_options(forcing_eta0=0.0))
    dt = time.time() - t0
    print("sec %.4f LM it %d pcg %d" % (dt, summ.num_iterations, tot), flush=True)
