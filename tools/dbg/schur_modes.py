"""the three forms of the pair-plan Schur kernel side by side (debug build: STBA_LIB=tmp_libs/dbg.so): device time at C5 and whether S is bitwise reproducible
0: slots dealt to waves by landmark range | 1: one list, waves add in turn (token) | 2: one list, arrival order (round 4)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
st = importlib.import_module("slam-tricks_amd")
class A: cams = 1000; pts = 100000; obs_per_pt = 10
s = bench.load_scene(A, 0)
rng = np.random.default_rng(1)
dc = rng.uniform(0.01, 0.1, (1000, 6)); dp = rng.uniform(0.01, 0.1, (len(s["pts0"]), 3))
ref = None
for mode, runs in ((2, 1), (3, 0), (3, 1), (2, 1), (3, 0), (3, 1)):
    os.environ["STBA_SCHUR_PLAN"] = str(mode); os.environ["STBA_SCHUR_RUNS"] = str(runs)
    eng = st.BAEngine(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
    eng.evaluate(jac=False); eng.normal_blocks()
    S0, r0 = eng.reduced_system(dc, dp)
    same = True
    for _ in range(4):
        S1, r1 = eng.reduced_system(dc, dp)
        same = same and np.array_equal(np.tril(S1), np.tril(S0)) and np.array_equal(r1, r0)
    if ref is None: ref = np.tril(S0)
    dev = np.abs(np.tril(S0) - ref).max() / np.abs(ref).max()
    eng.lm_iterations(2)
    ms, atomics, pairs = eng.time_schur(20)
    print(f"mode {mode} runs {runs}: schur {ms:.4f} ms  reproducible {same}  max rel dev from the first form {dev:.2e}", flush=True)
    eng.close()
