"""Factorisation time under scheduling-model knobs (debug build: STBA_DEBUG_KNOBS=1 at build time).
usage: python tools/chol_knobs.py n "VAR=a:VAR2=b" ...   -- one child process per configuration (the plan is built once per process)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import importlib, sys
sys.path.insert(0, %r)
st = importlib.import_module("slam-tricks_amd")
f, b = st.cholesky_time_split(%d, reps=10)
print("RESULT factor %%.4f ms backward %%.4f ms model %%.4f ms" %% (f, b, st.cholesky_schedule_model(%d) / 1e3))
"""
n = int(sys.argv[1])
for cfg in [""] + sys.argv[2:]:
    e = dict(os.environ)
    if cfg: e.update(dict(x.split("=", 1) for x in cfg.split(":")))
    p = subprocess.run([sys.executable, "-c", CHILD % (ROOT, n, n)], env=e, capture_output=True, text=True, timeout=600)
    res = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
    print(f"{cfg or 'default':60s} {res[0] if res else 'FAILED ' + p.stderr[-200:]}", flush=True)
