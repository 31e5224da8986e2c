"""one rank's share of the landmark-heavy scene at N ranks (landmark shard 0 of N): Schur step by task size (debug build, STBA_SCHUR_TASK_PAIRS)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
st = importlib.import_module("slam-tricks_amd"); sharding = importlib.import_module("slam-tricks_amd.sharding")
class A: second_cams = 100; second_pts = 1000000
s = bench.load_second_scene(A, 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sh = sharding.make_shard(s, 0, N)
e = st.BAEngine(sh["cams0"], sh["pts0"], sh["obs_cam"], sh["obs_pt"], sh["obs_feat"], sh["cam_fixed"])
e.lm_iterations(2)
ms, at, pr = e.time_schur(10)
summ, _ = e.lm_iterations(10, phase_timing=1)
print(f"N {N} task pairs <= {os.environ.get('STBA_SCHUR_TASK_PAIRS', 'rule')}: schur kernel {ms:.3f} ms, pairs {pr:.3g}; phases per it: lin {summ.ms_linearize/10:.3f} schur {summ.ms_schur/10:.3f} solve {summ.ms_solve/10:.3f} backsub {summ.ms_backsub/10:.3f} cost {summ.ms_cost/10:.3f}", flush=True)
