"""offline sweep of the scheduling model behind the persistent Cholesky kernel (no GPU):
usage: python tools/sim_sweep.py N "VAR=a" "VAR=b VAR2=c" ...  -> predicted makespan per environment setting"""
import os, subprocess, sys
n = sys.argv[1]
code = "import importlib,sys; st=importlib.import_module('slam-tricks_amd'); print(st.cholesky_schedule_model(int(sys.argv[1])))"
for kv in sys.argv[2:]:
    env = dict(os.environ)
    for item in kv.split():
        k, v = item.split("=", 1)
        env[k] = v
    out = subprocess.run([sys.executable, "-c", code, n], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    print(f"{kv:50s} {out.stdout.strip()} {out.stderr.strip()[-200:]}")
