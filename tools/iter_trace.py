"""prints GPU idle gaps inside one LM iteration of the C5 bench (run under rocprofv3 --kernel-trace)"""
import csv, glob, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "run":
    import numpy as np
    st = importlib.import_module("slam-tricks_amd")
    scenes = importlib.import_module("slam-tricks_amd.scenes")
    cache = "/tmp/stba_scene_c1000_p100000_m10_s20.npz"
    if os.path.exists(cache):
        z = np.load(cache); s = {k: z[k] for k in z.files}
    else:
        s = scenes.st20_scene(n_cams=1000, n_pts=100000, max_obs_per_pt=10, seed=20, pix_noise=1e-3); np.savez(cache, **s)
    e = st.BAEngine(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
    e.lm_iterations(2)
    summ, _ = e.lm_iterations(4)
    print("ms/iter wall", summ.seconds_total * 1e3 / 4)
else:
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    lin = [i for i, r in enumerate(rows) if "ba_linearize_kernel<true, true>" in r["Kernel_Name"]]
    a, b = lin[-3], lin[-2]       # one full iteration between two linearisations
    t0 = int(rows[a]["Start_Timestamp"])
    print("iteration span us", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, "kernels", b - a)
    busy = 0; last_end = t0; gaps = []
    for r in rows[a:b]:
        s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s_ > last_end + 5000:
            gaps.append(((s_ - last_end) / 1e3, (last_end - t0) / 1e3, r["Kernel_Name"].split("(")[0][-40:]))
        last_end = max(last_end, e_)
    print("total gap us", sum(g[0] for g in gaps), "n gaps>5us", len(gaps))
    for g in sorted(gaps, reverse=True)[:12]:
        print("gap %.1f us at t=%.1f before %s" % g)
