"""Wall-clock of the DROP-IN path: the reference's call sites re-typed on include/stba/ceres.h (tests/cpp/test_ceres_shim.cpp),
built with g++ and timed as the reference times them (construction + Solve: st20-g2o/src/include/test_ceres.h:103-104,149,
st17-ceres/src/include/solver.hpp:253-288).  bench.py calls run_ba() / run_pnp(); standalone: python tools/drop_in_time.py [--c5]"""
import importlib
import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "slam-tricks_amd")
SRC = os.path.join(ROOT, "tests", "cpp", "test_ceres_shim.cpp")
# st17-ceres/img/release.png, hardware unstated, one thread, the timer spans construction + Solve
PUBLISHED_MS = {"pnp_dyn": 0.22288, "pnp_auto": 0.13804, "pnp_sized": 0.12422, "self_gauss_newton": 0.01920}


def build_exe(out_dir):
    exe = os.path.join(out_dir, "drop_in_shim")
    subprocess.check_call(["g++", "-std=c++17", "-O3", "-I", os.path.join(ROOT, "include"), SRC, "-L", PKG, "-lstba",
                           f"-Wl,-rpath,{PKG}", "-o", exe])
    return exe


def write_scene(path, s):
    with open(path, "wb") as f:
        f.write(struct.pack("iii", len(s["cams0"]), len(s["pts0"]), len(s["obs_cam"])))
        for key, dt in (("cams0", np.float64), ("pts0", np.float64), ("obs_cam", np.int32), ("obs_pt", np.int32), ("obs_feat", np.float64)):
            f.write(np.ascontiguousarray(s[key], dt).tobytes())
        f.write(np.ascontiguousarray(s["cam_fixed"][:, 0], np.uint8).tobytes())


def write_pnp(path, s):
    with open(path, "wb") as f:
        f.write(struct.pack("i", len(s["pts"])))
        f.write(np.ascontiguousarray(s["pose_true"], np.float64).tobytes())
        f.write(np.ascontiguousarray(s["pose_init"], np.float64).tobytes())
        f.write(np.ascontiguousarray(np.hstack([s["pts"], s["feats"]]), np.float64).tobytes())


def _lines(exe, *args, timeout=900):
    p = subprocess.run([exe, *args], capture_output=True, text=True, timeout=timeout)
    if p.returncode != 0:
        raise RuntimeError(f"{args}: rc {p.returncode}: {p.stderr[-400:]}")
    out = {}
    for line in p.stdout.splitlines():
        k, _, v = line.partition(" ")
        out[k] = v
    return out


def run_ba(exe, scene, tmp, reps=5, max_iterations=50, threads=1):
    """median over `reps` of the re-typed SolveWithCeresDynamicAutoDiff (test_ceres.h:98-152) on `scene`, seconds per phase"""
    f = os.path.join(tmp, "ba_scene.bin")
    write_scene(f, scene)
    out = _lines(exe, "time_ba", f, str(max_iterations), str(reps), str(threads))
    rows = []
    for r in range(reps):
        toks = out[f"time_ba_{r}"].split()
        d = {toks[i]: toks[i + 1] for i in range(0, len(toks), 2)}
        rows.append(d)
    keys = ["build", "solve", "destroy", "recognise", "pack", "engine_create", "device_solve", "write_back", "verify", "resolve", "minimizer"]
    med = {k: float(np.median([float(r[k]) for r in rows])) for k in keys}
    cams = np.array([float(x) for x in out["time_ba_cams"].split()]).reshape(-1, 7)
    return {"execution_path": rows[-1]["path"], "termination": int(rows[-1]["term"]), "iterations": int(rows[-1]["iters"]),
            "initial_cost": float(rows[-1]["initial"]), "final_cost": float(rows[-1]["final"]), "reps": reps, "host_threads": threads,
            "seconds": med, "solve_seconds_all": [float(r["solve"]) for r in rows]}, cams


def run_pnp(exe, pnp, tmp, reps=200):
    f = os.path.join(tmp, "pnp.bin")
    write_pnp(f, pnp)
    out = _lines(exe, "time_pnp", f, str(reps))
    res = {}
    for name in ("pnp_dyn", "pnp_auto", "pnp_sized"):
        ms = np.array([float(x) for x in out[name + "_ms"].split()])
        toks = out[name].split()
        d = {toks[i]: toks[i + 1] for i in range(0, len(toks), 2)}
        pose = np.array([float(x) for x in out[name + "_pose"].split()])
        res[name] = {"ms_median": float(np.median(ms[1:])), "ms_min": float(ms[1:].min()), "ms_first": float(ms[0]), "reps": reps,
                     "iterations": int(d["iters"]), "final_cost": float(d["final"]), "termination": int(d["term"]),
                     "execution_path": d["path"], "published_ms": PUBLISHED_MS[name], "pose": pose.tolist()}
    return res


if __name__ == "__main__":
    scenes = importlib.import_module("slam-tricks_amd.scenes")
    with tempfile.TemporaryDirectory() as tmp:
        exe = build_exe(tmp)
        pnp = scenes.pnp_scene(seed=17)
        print(json.dumps({"published_workload": run_pnp(exe, pnp, tmp)}))
        if "--c5" in sys.argv:
            s = scenes.st20_scene(n_cams=1000, n_pts=100000, max_obs_per_pt=10, seed=20, pix_noise=1e-3)
        else:
            s = scenes.st20_scene()
        d, _ = run_ba(exe, s, tmp)
        print(json.dumps({"drop_in": d}))
