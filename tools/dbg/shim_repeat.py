"""every call site of the re-typed reference program (tests/cpp/test_ceres_shim.cpp: bounds demo, PnP four ways, the BA call site with the
built-in and the user's factor, pose graph) N times over: does every run print the same numbers (timing fields aside)?
usage: python tools/dbg/shim_repeat.py [N]"""
import hashlib, importlib, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import drop_in_time as D
scenes = importlib.import_module("slam-tricks_amd.scenes")
import test_cpp_shim as T
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
pnp = scenes.pnp_scene(seed=17); sc = scenes.st20_scene()
D.write_pnp("/tmp/pnp.bin", pnp); D.write_scene("/tmp/sc.bin", sc)
pg = scenes.pose_graph_scene(n_nodes=400, loops_per_node=3, seed=4, sigma_t=0.02, sigma_r=0.004, turns=6)
T.write_pg("/tmp/pg.bin", pg)
exe = D.build_exe("/tmp")
drop = re.compile(r"(seconds|secs|time|ms|wall)\s+\S+")
for name, args in (("call sites", ["/tmp/pnp.bin", "/tmp/sc.bin", "/tmp/sc.bin"]), ("pose graph", ["pg", "/tmp/pg.bin"])):
    seen = {}
    for k in range(N):
        p = subprocess.run([exe, *args], capture_output=True, text=True)
        if p.returncode != 0: print("rc", p.returncode, p.stderr[-300:]); continue
        lines = [drop.sub("", l) for l in p.stdout.splitlines() if not re.search(r"secs|seconds|_time|wall", l.split(" ")[0])]
        h = hashlib.md5("\n".join(lines).encode()).hexdigest()
        if h not in seen and seen:
            ref = seen[next(iter(seen))][1]
            diff = [(a, b) for a, b in zip(ref, lines) if a != b][:3]
            print(name, "run", k, "differs:", [(a[:120], b[:120]) for a, b in diff], flush=True)
        seen.setdefault(h, [0, lines])[0] += 1
    print(name, {h: v[0] for h, v in seen.items()}, flush=True)
