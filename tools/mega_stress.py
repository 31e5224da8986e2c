"""determinism stress test of the persistent Cholesky kernel: the task graph fixes the order of every
floating-point operation, so repeated factorisations of the same matrix must agree BIT FOR BIT; any
difference is a synchronisation / cache-coherence race.
usage: python tools/mega_stress.py [n] [reps] [threads]"""
import importlib, os, sys, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
st = importlib.import_module("slam-tricks_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nthreads = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rng = np.random.default_rng(5)
B = rng.standard_normal((n, n // 2))
A = B @ B.T + n * 0.01 * np.eye(n)
b = rng.standard_normal(n)
Lref = np.linalg.cholesky(A)
bad = [0] * nthreads
def run(tid):
    x0 = None
    for r in range(reps):
        try:
            x = st.cholesky_factor(A)
        except Exception as e:
            bad[tid] += 1
            print(tid, "rep", r, "FAILED:", e, flush=True)
            continue
        if x0 is None:
            x0 = x
            print(tid, "rel err of L vs numpy", np.abs(x - Lref).max() / np.abs(Lref).max(), flush=True)
        elif not np.array_equal(x, x0):
            bad[tid] += 1
            print(tid, "rep", r, "DIFFERS: max rel", np.abs(x - x0).max() / np.abs(x0).max(), flush=True)
th = [threading.Thread(target=run, args=(i,)) for i in range(nthreads)]
[t.start() for t in th]; [t.join() for t in th]
print("mismatching repetitions:", bad)
sys.exit(1 if any(bad) else 0)
