"""prints the kernel timeline of one look-ahead Cholesky (run under rocprofv3 --kernel-trace)"""
import csv, glob, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "run":
    st = importlib.import_module("slam-tricks_amd")
    print("chol ms", st.cholesky_time(6000, reps=2))
else:
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows = [r for r in rows if "chol_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # last factorisation only
    print(list(rows[0].keys()))
    diags = [i for i, r in enumerate(rows) if "chol_diag" in r["Kernel_Name"]]
    i0 = diags[-36] - 2
    t0 = int(rows[i0]["Start_Timestamp"])
    for r in rows[i0:i0 + 24]:
        name = r["Kernel_Name"].split("(")[0].replace("stba::", "")
        print(f'{name:28s} grid={r.get("Grid_Size", r.get("Grid_Size_X")):>8s} q={r.get("Queue_Id","?"):>3s} start={(int(r["Start_Timestamp"])-t0)/1e3:9.1f}us end={(int(r["End_Timestamp"])-t0)/1e3:9.1f}us dur={(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:7.1f}')
