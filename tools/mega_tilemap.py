"""per-tile error map of one factorisation (debugging aid). usage: python tools/mega_tilemap.py [n] [tries]"""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
st = importlib.import_module("slam-tricks_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
tries = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = np.random.default_rng(5)
B = rng.standard_normal((n, n // 2))
A = B @ B.T + n * 0.01 * np.eye(n)
Lref = np.linalg.cholesky(A)
L = st.lib()
for k in range(tries):
    X = A.copy()
    rc = L.stba_cholesky_factor(X.ctypes.data_as(C.POINTER(C.c_double)), n, C.c_void_p(0))
    X = np.tril(X)
    nb = (n + 127) // 128
    print("try", k, "status", rc)
    scale = np.abs(Lref).max()
    for i in range(nb):
        row = ""
        for j in range(i + 1):
            e = np.abs(X[128*i:128*i+128, 128*j:128*j+128] - Lref[128*i:128*i+128, 128*j:128*j+128])
            e = np.nan_to_num(e, nan=1e9).max() / scale
            row += "." if e < 1e-10 else ("x" if e < 1e-3 else "X")
        print(f"{i:3d} {row}")
