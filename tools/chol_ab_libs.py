"""A/B of several BUILDS of the library (tmp_libs/<name>.so, tools/build_dbg.sh), interleaved to ride out box drift:
usage: python tools/chol_ab_libs.py n rounds nameA nameB ...   -> median / min per build; first a determinism + accuracy check of each"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import importlib, sys
sys.path.insert(0, %r)
st = importlib.import_module("slam-tricks_amd")
st.cholesky_time_split(%d, reps=3)
f, b = st.cholesky_time_split(%d, reps=20)
print("RESULT %%.5f %%.5f" %% (f, b))
"""
CHECK = r"""
import importlib, sys
import numpy as np
sys.path.insert(0, %r)
st = importlib.import_module("slam-tricks_amd")
n = %d
rng = np.random.default_rng(5)
B = rng.standard_normal((n, n // 2)); A = B @ B.T + n * 0.01 * np.eye(n)
L0 = st.cholesky_factor(A)
same = all(np.array_equal(st.cholesky_factor(A), L0) for _ in range(4))
Lr = np.linalg.cholesky(A)
print("CHECK deterministic", same, "rel err", np.abs(L0 - Lr).max() / np.abs(Lr).max(), "timeouts", st.cholesky_timeout_count())
"""
n, rounds = int(sys.argv[1]), int(sys.argv[2])
libs = sys.argv[3:]
def run(code, lib):
    e = dict(os.environ); e["STBA_LIB"] = os.path.join(ROOT, "tmp_libs", lib + ".so")
    return subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900)
for l in libs:
    p = run(CHECK % (ROOT, min(n, int(os.environ.get('CHECK_N', '6000')))), l)   # CHECK_N: check at a larger size (numpy factors it on the host)
    print(l, (p.stdout.strip().splitlines() or [p.stderr[-300:]])[-1], flush=True)
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        p = run(CHILD % (ROOT, n, n), l)
        x = [y for y in p.stdout.splitlines() if y.startswith("RESULT")]
        if x: res[l].append((float(x[0].split()[1]), float(x[0].split()[2])))
for l in libs:
    v = np.array([r[0] for r in res[l]]); w = np.array([r[1] for r in res[l]])
    print(f"{l:20s} factor median {np.median(v):.4f} min {v.min():.4f} max {v.max():.4f} ms | backward median {np.median(w):.4f} | sum {np.median(v + w):.4f} ({len(v)} runs)", flush=True)
