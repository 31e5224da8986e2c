"""the landmark-heavy scene's Schur step by task size (debug build): STBA_LIB=tmp_libs/dbg.so python tools/dbg/lh_schur.py"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
st = importlib.import_module("slam-tricks_amd")
class A: second_cams = 100; second_pts = 1000000
s = bench.load_second_scene(A, 0)
for tp in (int(os.environ.get("STBA_SCHUR_TASK_PAIRS", 1 << 30)),):
    e = st.BAEngine(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
    e.lm_iterations(2)
    ms, at, pr = e.time_schur(5)
    summ, _ = e.lm_iterations(5)
    print(f"task pairs <= {tp}: schur {ms:.3f} ms, pairs {pr:.3g}, cost after 7 its {summ.final_cost:.10e}", flush=True)
    e.close()
