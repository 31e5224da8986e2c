"""kernel timeline of the last LM iterations of a C4 solve (run under rocprofv3 --kernel-trace): python tools/c4_iter_trace.py run | <dir>"""
import csv, glob, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "run":
    st = importlib.import_module("slam-tricks_amd")
    scenes = importlib.import_module("slam-tricks_amd.scenes")
    s = scenes.pose_graph_scene(n_nodes=10000, loops_per_node=3, seed=4)
    for _ in range(2):
        e = st.PGEngine(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
        summ, tr, tot = e.solve()
    print("ms/solve", summ.seconds_total * 1e3, summ.num_iterations, tot)
else:
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    pc = [i for i, r in enumerate(rows) if "pg_pcg_persistent_kernel" in r["Kernel_Name"]]
    a, b = pc[-4], pc[-2]          # two full LM iterations of the lagged regime (PCG start to PCG start)
    t0 = int(rows[a]["Start_Timestamp"])
    print("span of 2 iterations us", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3)
    for r in rows[a:b + 1]:
        name = r["Kernel_Name"].split("(")[0].replace("stba::", "").replace("(anonymous namespace)::", "").replace("void ", "")
        print(f'{name[:40]:40s} q={r.get("Queue_Id", "?"):>3s} start={(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} dur={(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f}')
