"""A/B of factorisation times under scheduling knobs (debug build), interleaved to ride out box drift:
usage: python tools/chol_ab.py n rounds "VAR=a:VAR2=b" "VAR=c" ...   -> median / min per configuration"""
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
n, rounds = int(sys.argv[1]), int(sys.argv[2])
cfgs = sys.argv[3:]
res = {c: [] for c in cfgs}
for r in range(rounds):
    for c in cfgs:
        e = dict(os.environ)
        if c and c != "-": e.update(dict(x.split("=", 1) for x in c.split(":")))
        p = subprocess.run([sys.executable, "-c", CHILD % (ROOT, n, n)], env=e, capture_output=True, text=True, timeout=600)
        l = [x for x in p.stdout.splitlines() if x.startswith("RESULT")]
        if l: res[c].append(float(l[0].split()[1]))
for c in cfgs:
    v = np.array(res[c])
    print(f"{c:50s} median {np.median(v):.4f} min {v.min():.4f} max {v.max():.4f} ms ({len(v)} runs)", flush=True)
