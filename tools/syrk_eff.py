"""per-launch efficiency of the trailing-update kernel (run under rocprofv3 --kernel-trace)"""
import csv, glob, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    import numpy as np, ctypes as C
    st = importlib.import_module("slam-tricks_amd")
    print(st.cholesky_profile(6000))
else:
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "chol_syrk" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-46:]      # the profiled (serial) factorisation is the last one
    tot = 0; ideal_tot = 0
    for r in rows:
        wg = int(r["Grid_Size_X"]) // 256
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        flops = wg * 2.0 * 64 * 128 * 128
        ideal = flops / 78.6e12 * 1e6
        tot += dur; ideal_tot += ideal
        print(f"wgs {wg:5d} dur {dur:7.1f} us ideal {ideal:6.1f} eff {ideal/dur:5.2f} per-CU-rounds {wg/768:5.2f}")
    print("total", tot, "ideal", ideal_tot)
