"""correctness / determinism of the factorisation for one build: STBA_LIB=tmp_libs/<name>.so python tools/dbg/chol_check.py [n ...]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
st = importlib.import_module("slam-tricks_amd")
for n in [int(a) for a in sys.argv[1:]] or [127, 1000, 6000]:
    rng = np.random.default_rng(5)
    B = rng.standard_normal((n, max(4, n // 2))); A = B @ B.T + n * 0.01 * np.eye(n)
    L0 = st.cholesky_factor(A)
    same = all(np.array_equal(st.cholesky_factor(A), L0) for _ in range(4))
    Lr = np.linalg.cholesky(A)
    E = np.abs(L0 - Lr)
    w = np.unravel_index(np.argmax(E), E.shape)
    print(os.environ.get("STBA_LIB", "product"), os.environ.get("STBA_MEGA_GRID_PER_CU", ""), "n", n, "deterministic", same, "rel err", E.max() / np.abs(Lr).max(), "worst at", w,
          "timeouts", st.cholesky_timeout_count(), flush=True)
