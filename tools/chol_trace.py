"""prints the kernel timeline of the last look-ahead Cholesky of a run under rocprofv3 --kernel-trace
usage:  rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/chol_trace.py run [n]
        python tools/chol_trace.py DIR [n]"""
import csv, glob, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
if sys.argv[1] == "run":
    st = importlib.import_module("slam-tricks_amd")
    print("chol ms", st.cholesky_time(n, reps=3))
else:
    nblk = (n + 1 + 127) // 128
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows = [r for r in rows if "chol_" in r["Kernel_Name"] and "pad" not in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    diags = [i for i, r in enumerate(rows) if "chol_diag" in r["Kernel_Name"]]
    i0 = diags[-nblk]
    t0 = int(rows[i0]["Start_Timestamp"])
    short = {"chol_diag_kernel": "D", "chol_trsm_kernel": "T", "chol_syrk_kernel<64>": "U", "chol_syrk_kernel<128>": "U", "chol_bwd_step_kernel": "B"}
    for r in rows[i0:]:
        name = r["Kernel_Name"].split("(")[0].replace("stba::", "").replace("void ", "")
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        grid = r.get("Grid_Size", r.get("Grid_Size_X"))
        wg = r.get("Workgroup_Size", r.get("Workgroup_Size_X"))
        print(f'{short.get(name, name):2s} q={r.get("Queue_Id","?"):>2s} wgs={int(grid)//int(wg):5d} start={s:9.1f} end={e:9.1f} dur={e-s:7.1f}')
