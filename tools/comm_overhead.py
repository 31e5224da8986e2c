"""What the several-ranks code path costs next to the one-rank path, on ONE GPU: the C5 LM iteration with a 1-rank RCCL
communicator attached (pack -> ncclAllReduce -> unpack, the reduced finalisation, the collective of the trial block) against
the same engine without one.  usage: python tools/comm_overhead.py [steps]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
st = importlib.import_module("slam-tricks_amd")
class A: cams = 1000; pts = 100000; obs_per_pt = 10
s = bench.load_scene(A, 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
def run(with_comm):
    eng = st.BAEngine(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
    comm = None
    if with_comm:
        comm = st.Comm(st.comm_unique_id(), 0, 1)
        eng.set_comm(comm)
    eng.lm_iterations(3)
    best = 1e9
    for _ in range(3):
        eng.set_params(s["cams0"], s["pts0"])
        t0 = time.perf_counter()
        summ, tr = eng.lm_iterations(steps)
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    out = (best, summ.final_cost, summ.allreduce_bytes / max(1, summ.allreduce_calls))
    eng.close()
    if comm: comm.close()
    return out
a = run(False); b = run(True)
print(f"one-rank path {a[0]:.4f} ms/step (final cost {a[1]:.12e}); with a 1-rank communicator {b[0]:.4f} ms/step (final cost {b[1]:.12e}, "
      f"{b[2] / 1e6:.1f} MB per all-reduce): +{b[0] - a[0]:.4f} ms")
