"""factorisation time (hipEvent, persistent kernel only) against the scheduling model for several system sizes
usage: python tools/size_sweep.py [n ...]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
st = importlib.import_module("slam-tricks_amd")
for n in [int(a) for a in sys.argv[1:]] or [3000, 6000, 9000, 12000, 16000]:
    ms_f, ms_b = st.cholesky_time_split(n, reps=5)
    model = st.cholesky_schedule_model(n) / 1e3
    flops = n ** 3 / 3.0 + n ** 2 / 2.0
    print(f"n = {n:6d}: factorisation {ms_f:8.3f} ms (model {model:8.3f} ms), {flops / ms_f / 1e9:6.2f} TFLOP/s FP64, backward {ms_b:6.3f} ms", flush=True)
