#!/usr/bin/env python
"""bench.py -- LM iterations/s + residuals/s on the 1k-camera / 100k-point synthetic BA (BASELINE
config C5: 1 000 cameras, 100 000 landmarks, 1 000 000 reprojection observations, FP64).

A "step" is ONE Levenberg-Marquardt iteration of the hot path over the whole problem:
residual+Jacobian kernel -> block J^T J / J^T r -> landmark Schur complement -> dense FP64
Cholesky of the 6000 x 6000 reduced camera system (MFMA) -> back-substitution -> manifold
update -> residual kernel at the trial point (stba_ba_lm_iterations: every iteration does all
of that, accepted or not).  Inputs are resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--reps R] [--cams C --pts P --obs-per-pt M]

Protocol: W untimed warm-up steps, then R repetitions of EXACTLY K timed steps, each repetition
from the same start point and bracketed by barrier + device synchronisation, max over ranks; the JSON line
reports the MEDIAN repetition (`value`, `ms_per_step`) and lists all of them.  A speed-up over the CPU leg is
printed only after the GPU leg and the CPU leg, run for the same fixed iteration count from the same start,
agree in final cost (1e-6 relative) and camera poses (1e-5).

N > 1: one process per GPU.  `python bench.py --gpus N` re-launches itself under torch.distributed.run
(--nproc-per-node N, 127.0.0.1); when the driver already launched N ranks (WORLD_SIZE set) it must equal
--gpus.  Landmarks are sharded across ranks, cameras replicated; the cross-rank sum of the reduced camera
system is a native RCCL all-reduce (slam-tricks_amd.Comm: ncclAllReduce on the engine's stream; the 128-byte
id travels over torch.distributed).  Strong scaling: the problem is fixed.  Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix peak (SURVEY.md 8d / AMD datasheet)
FP64_MFMA_MEASURED_CEILING_TFLOPS = 77.3   # profiles/mfma_f64_microbench.txt (pure-MFMA loop, 512-thread blocks, this chip)
BYTES_PER_OBS_JAC = 106.5    # residual + COMPACT Jacobian kernel at 10 obs/landmark: read 24 B (feature, 2 indices) + 24 B / 10 (landmark)
                             # + 0.06 B (camera table); write 16 B (r) + 64 B ({xn, yn, P 2x3}: the 2x6 | 2x3 form of SURVEY.md 8d, 186.5 B,
                             # carries 80 redundant bytes per observation -- DESIGN.md 4)


def load_scene(args, rank):
    scenes = importlib.import_module("slam-tricks_amd.scenes")
    dense = bool(getattr(args, "dense_visibility", False))
    tag = f"c{args.cams}_p{args.pts}_m{'all' if dense else args.obs_per_pt}_s20"
    cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"stba_scene_{tag}.npz")
    if os.path.exists(cache):
        try:
            z = np.load(cache)
            return {k: z[k] for k in z.files}
        except Exception:
            pass
    if dense:
        s = scenes.st20_scene(n_cams=args.cams, n_pts=args.pts, max_obs_per_pt=None, seed=20, pix_noise=1e-3, half_w=3.0, half_h=3.0,
                              retriangulate=False)
    else:
        s = scenes.st20_scene(n_cams=args.cams, n_pts=args.pts, max_obs_per_pt=args.obs_per_pt, seed=20,
                              pix_noise=1e-3)
    if rank == 0:
        try:
            np.savez(cache + f".tmp{os.getpid()}.npz", **s)
            os.replace(cache + f".tmp{os.getpid()}.npz", cache)
        except Exception:
            pass
    return s


def make_engine(st, sharding, torch, dist, sh, stream, rank, world, local_rank, have_gpu, args):
    """this rank's engine on `stream`, with the cross-rank sum wired up: (engine, native communicator or None, description)"""
    eng = st.BAEngine(sh["cams0"], sh["pts0"], sh["obs_cam"], sh["obs_pt"], sh["obs_feat"], sh["cam_fixed"], stream=stream)
    collective = "none"
    comm = None
    if world > 1:
        if args.hook == "native":
            # the decision native communicator | torch hook is COLLECTIVE: rank 0 always broadcasts (the id or None), and after
            # the communicator is created the ranks agree on whether every one of them succeeded -- a rank that fell back
            # alone would leave the others blocked in ncclCommInitRank or issue mismatched collectives
            uid = None
            if rank == 0:
                try:
                    uid = st.comm_unique_id()
                except Exception as e:      # noqa: BLE001
                    print(f"[rank 0] native communicator: no unique id ({e!r})", file=sys.stderr, flush=True)
            box = [uid]
            dist.broadcast_object_list(box, src=0)
            ok = 0.0
            if box[0] is not None:
                try:
                    comm = st.Comm(box[0], rank, world, device=local_rank)
                    ok = 1.0
                except Exception as e:      # noqa: BLE001
                    print(f"[rank {rank}] native communicator failed ({e!r})", file=sys.stderr, flush=True)
                    comm = None
            agree = torch.tensor([ok], dtype=torch.float64, device="cuda" if (have_gpu and args.backend == "nccl") else "cpu")
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)
            if float(agree.item()) >= 1.0:
                eng.set_comm(comm)
                collective = "native RCCL (ncclAllReduce on the engine stream, stba_comm)"
            else:
                if comm is not None:
                    comm.close()
                comm = None
                if rank == 0:
                    print("native communicator not available on every rank: all ranks use the torch hook", file=sys.stderr, flush=True)
        if comm is None:
            eng.set_allreduce(sharding.torch_allreduce_hook(dist, torch), rank, world)
            collective = "torch.distributed all_reduce (RCCL) through the Python hook"
    return eng, comm, collective


def ranks_hold_identical_cameras(eng, torch, dist, world, have_gpu, args):
    """every rank factors the same reduced system, so the camera blocks must be BIT-identical everywhere: compared in-run
    (a 64-bit digest of the camera array per rank, gathered)"""
    import hashlib
    cams, _ = eng.get_params()
    dig = int.from_bytes(hashlib.sha1(np.ascontiguousarray(cams).tobytes()).digest()[:7], "little")
    if world == 1:
        return True, [dig]
    dev = "cuda" if (have_gpu and args.backend == "nccl") else "cpu"
    mine = torch.tensor([dig], dtype=torch.int64, device=dev)
    allv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    vals = [int(v.item()) for v in allv]
    same = all(v == vals[0] for v in vals)
    # what a `false` would mean: how far apart the ranks are, and whether a rank's persistent factorisation gave up (a time-out sends it
    # to the stage kernels, whose order of operations -- and last bits -- differ: seen when several processes share ONE device)
    global LAST_RANK_CHECK
    st = importlib.import_module("slam-tricks_amd")
    mc = torch.from_numpy(np.ascontiguousarray(cams)).to(dev)
    ref = mc.clone()
    dist.broadcast(ref, src=0)
    far = torch.tensor([float((mc - ref).abs().max()), float(st.cholesky_timeout_count())], dtype=torch.float64, device=dev)
    allf = [torch.zeros_like(far) for _ in range(world)]
    dist.all_gather(allf, far)
    LAST_RANK_CHECK = {"max_abs_diff_to_rank0": max(float(v[0]) for v in allf), "factorisation_timeouts_per_rank": [int(v[1]) for v in allf]}
    return same, vals


LAST_RANK_CHECK = None


def predicted_scaling(ph, ms_step, allreduce_ms, ar_bytes, world, note):
    """what landmark sharding can and cannot buy, from ONE run's phase times (a model, printed so that a measured N-GPU point can be
    held against it): factorisation + backward substitution replicated on every rank, Jacobian / Schur / back-substitution /
    trial-cost work divided by N, one packed all-reduce of the reduced system per build added (bounds: a ring over one xGMI link,
    48 GB/s each way; reduce-scatter + all-gather over all N - 1 links)"""
    t_repl = ph["ms_solve"]
    t_shard = max(0.0, ph["ms_linearize"] + ph["ms_schur"] - allreduce_ms + ph["ms_backsub"] + ph["ms_cost"]) * world
    t_other = max(0.0, ms_step - (ph["ms_linearize"] + ph["ms_schur"] + ph["ms_solve"] + ph["ms_backsub"] + ph["ms_cost"]))
    pred = {}
    for N in (1, 2, 4, 8):
        ring = 0.0 if N == 1 else 2.0 * (N - 1) / N * ar_bytes / 48e9 * 1e3
        direct = 0.0 if N == 1 else 2.0 * ar_bytes / N / 48e9 * 1e3
        lo, hi = t_repl + t_other + t_shard / N + direct, t_repl + t_other + t_shard / N + ring
        pred[str(N)] = {"ms_per_step_best": lo, "ms_per_step_ring": hi, "it_per_s_best": 1e3 / lo, "it_per_s_ring": 1e3 / hi}
    base = pred["1"]["ms_per_step_best"]
    return {"model": "T(N) = replicated (factorisation + backward substitution) + other + sharded / N + all-reduce(N); terms from this run's "
                     "phase_ms_per_step; all-reduce of the packed reduced system: ring over one 48 GB/s xGMI link | direct over N - 1 links",
            "replicated_ms": t_repl, "sharded_ms_total": t_shard, "other_ms": t_other, "allreduce_bytes": ar_bytes,
            "per_gpus": pred, "speedup_at_8_best": base / pred["8"]["ms_per_step_best"],
            "speedup_at_8_ring": base / pred["8"]["ms_per_step_ring"], "note": note}


def time_scene(st, sharding, torch, dist, s, args, rank, world, local_rank, have_gpu, steps, reps, warmup):
    """the bench protocol on one scene: shard, engine + collective, warm-up, R x K timed LM iterations (barrier + synchronise on both
    sides, max over ranks, median over repetitions), one instrumented pass for the phase times, the in-run identity check"""
    sh = sharding.make_shard(s, rank, world)
    stream_obj = torch.cuda.Stream()
    eng, comm, collective = make_engine(st, sharding, torch, dist, sh, stream_obj.cuda_stream, rank, world, local_rank, have_gpu, args)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
    if warmup > 0:
        eng.lm_iterations(warmup)
    rep_ms, summ = [], None
    for _ in range(max(1, reps)):
        eng.set_params(sh["cams0"], sh["pts0"])
        sync()
        t0 = time.perf_counter()
        summ, _ = eng.lm_iterations(steps)
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if (have_gpu and args.backend == "nccl") else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        rep_ms.append(1e3 * dt / steps)
    same, digests = ranks_hold_identical_cameras(eng, torch, dist, world, have_gpu, args)
    ms_step = float(np.median(rep_ms))
    eng.set_params(sh["cams0"], sh["pts0"])
    sync()
    summ_i, _ = eng.lm_iterations(steps, phase_timing=1)
    sync()
    phase_keys = ("ms_linearize", "ms_schur", "ms_solve", "ms_backsub", "ms_cost", "ms_allreduce")
    phases = np.array([getattr(summ_i, k) / steps for k in phase_keys], dtype=np.float64)
    if world > 1:
        tp = torch.tensor(phases, dtype=torch.float64, device="cuda" if (have_gpu and args.backend == "nccl") else "cpu")
        dist.all_reduce(tp, op=dist.ReduceOp.MAX)
        phases = tp.cpu().numpy()
    n_cams, n_pts, n_obs = len(s["cams0"]), len(s["pts0"]), len(s["obs_cam"])
    ph = {k: float(v) for k, v in zip(phase_keys[:5], phases[:5])}
    ar_bytes = float(summ_i.allreduce_bytes / max(1, summ_i.allreduce_calls)) if world > 1 else 40.7e6 * (n_cams / 1000.0) ** 2
    local_counts = np.zeros(world, dtype=np.float64)
    local_counts[rank] = float(len(sh["obs_cam"]))
    if world > 1:
        tc = torch.tensor(local_counts, dtype=torch.float64, device="cuda" if (have_gpu and args.backend == "nccl") else "cpu")
        dist.all_reduce(tc)
        local_counts = tc.cpu().numpy()
    out = {"value": 1e3 / ms_step, "unit": "LM iterations/s", "residuals_per_sec": 1e3 / ms_step * 2.0 * n_obs, "ms_per_step": ms_step,
           "n_cams": n_cams, "n_pts": n_pts, "n_obs": n_obs, "observations_per_rank": [int(v) for v in local_counts],
           "reps_ms_per_step": rep_ms, "steps": steps, "n_gpus": world, "scaling": "strong",
           "config": {"workload": f"{n_cams} cams, {n_pts} pts, {n_obs} obs, Schur + dense {6 * n_cams}x{6 * n_cams} Cholesky, "
                                  "st20 spiral/cube scene seed 20, pixel noise 1e-3",
                      "collective": collective, "schur_form": "dense product" if eng.schur_mode() == eng.SCHUR_DENSE else "pair plan"},
           "phase_ms_per_step": ph, "allreduce_ms": float(phases[5]), "allreduce_bytes": ar_bytes,
           "camera_blocks_identical_on_all_ranks": bool(same), "rank_check": LAST_RANK_CHECK, "final_cost": summ.final_cost,
           "predicted_scaling": predicted_scaling(ph, ms_step, float(phases[5]), ar_bytes, world,
                                                  "few cameras, many landmarks: the replicated factorisation is small and everything else divides by N")}
    eng.close()
    if comm is not None:
        comm.close()
    return out


def load_second_scene(args, rank):
    """the landmark-heavy scene (few cameras, many landmarks, 10 observations per landmark); the landmarks start at their true positions
    (no re-triangulation pass: at 10^7 observations that alone took minutes), the cameras at the reference's noise"""
    scenes = importlib.import_module("slam-tricks_amd.scenes")
    tag = f"c{args.second_cams}_p{args.second_pts}_m10_s20_notri"
    cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"stba_scene_{tag}.npz")
    if os.path.exists(cache):
        try:
            z = np.load(cache)
            return {k: z[k] for k in z.files}
        except Exception:
            pass
    s = scenes.st20_scene(n_cams=args.second_cams, n_pts=args.second_pts, max_obs_per_pt=10, seed=20, pix_noise=1e-3, retriangulate=False)
    if rank == 0:
        try:
            np.savez(cache + f".tmp{os.getpid()}.npz", **s)
            os.replace(cache + f".tmp{os.getpid()}.npz", cache)
        except Exception:
            pass
    return s


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def relaunch(args):
    """`python bench.py --gpus N` without a launcher: one process per GPU under torch.distributed.run"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_quota():
    """CPUs this process may use at once: the cgroup's cpu.max quota (v2) / cfs quota (v1) and the affinity mask -- on the GPU boxes of
    this pool the container sees 256 logical cores and is throttled to 16 CPUs' worth of time (cpu.max = 1600000 100000,
    profiles/r6_host_cpu_quota.txt): OpenMP teams larger than that run SLOWER, which is what the thread sweep of round 5 showed"""
    q = None
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if a != "max":
            q = float(a) / float(b)
    except Exception:
        try:
            a = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            b = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if a > 0:
                q = a / b
        except Exception:
            pass
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = os.cpu_count() or 1
    return q, aff


def pose_err(a, b):
    dq = np.minimum(np.abs(a[:, :4] - b[:, :4]).max(1), np.abs(a[:, :4] + b[:, :4]).max(1)).max()
    return float(dq), float(np.abs(a[:, 4:] - b[:, 4:]).max())


def ceres_row(scene_file, n_iter):
    """opportunistic real-Ceres baseline: built only if a Ceres installation is found on this
    box (header + library), semantics of st20-g2o/src/include/test_ceres.h:98-152 (SPARSE_SCHUR, analytic
    cost function), num_threads 1 and all cores.  Never assumed: the image ships no Ceres."""
    inc = [d for d in ("/usr/include", "/usr/local/include", "/opt/ceres/include") if os.path.exists(os.path.join(d, "ceres", "ceres.h"))]
    if not inc:
        return {"found": False, "reason": "ceres/ceres.h not found in /usr/include, /usr/local/include, /opt/ceres/include"}
    src = os.path.join(ROOT, "tools", "ceres_baseline.cpp")
    exe = os.path.join(os.environ.get("TMPDIR", "/tmp"), "stba_ceres_baseline")
    eig = [d for d in ("/usr/include/eigen3", "/usr/local/include/eigen3") if os.path.isdir(d)]
    cmd = ["g++", "-O3", "-march=native", "-std=c++17", src, "-o", exe] + [f"-I{d}" for d in inc + eig] + ["-lceres", "-lglog", "-lpthread"]
    try:
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        rows = {}
        for th in (1, os.cpu_count() or 1):
            out = subprocess.run([exe, scene_file, str(n_iter), str(th)], capture_output=True, text=True, timeout=1800).stdout
            rows[str(th)] = json.loads(out.strip().splitlines()[-1])
        return {"found": True, "rows": rows}
    except Exception as e:      # noqa: BLE001
        return {"found": False, "reason": f"Ceres header present but the baseline did not build/run: {e!r}"}


# algorithmic HBM bytes per edge of the two pose-graph kernels (FP64):
#   residual + Jacobian: read 8 B (two indices) + 56 B (measurement) + 2 x 56 B poses / (edge ends per node = 2 m / n = 8);
#                        write 48 B (r) + 2 x 288 B (Ji, Jj)
#   matrix-free product, edge kernel: read 8 B + 2 x 288 B (Ji, Jj) + 2 x 48 B of p / 8; write 96 B (u = Ji^T t | Jj^T t, which the
#                        node kernel gathers: no atomics)
def drop_in_rows(s, eng, out):
    """OUTSIDE the timed region: the wall-clock of the path the reference would actually call.
    `drop_in`: ceres::Solve through include/stba/ceres.h for the re-typed SolveWithCeresDynamicAutoDiff (test_ceres.h:98-152) on THIS
    scene to convergence, one host thread (the reference pins num_threads = 1, test_ceres.h:143), broken into the phases the header
    reports (Solver::Summary::phases), next to the caller's own construction time and to the C ABI underneath.
    `published_workload`: the three st17 PnP call sites (solver.hpp:247-385) timed as the reference times them (construction +
    Solve, solver.hpp:253-288) beside BASELINE.md's published numbers."""
    import importlib.util as ilu
    import tempfile
    res = {}
    try:
        spec = ilu.spec_from_file_location("drop_in_time", os.path.join(ROOT, "tools", "drop_in_time.py"))
        dit = ilu.module_from_spec(spec)
        spec.loader.exec_module(dit)
        scenes = importlib.import_module("slam-tricks_amd.scenes")
        st = importlib.import_module("slam-tricks_amd")
        with tempfile.TemporaryDirectory() as tmp:
            exe = dit.build_exe(tmp)
            pw = dit.run_pnp(exe, scenes.pnp_scene(seed=17), tmp, reps=200)
            truth = scenes.pnp_scene(seed=17)["pose_true"]
            for v in pw.values():
                pose = np.array(v.pop("pose"))
                v["pose_error"] = float(max(min(np.abs(pose[:4] - truth[:4]).max(), np.abs(pose[:4] + truth[:4]).max()), np.abs(pose[4:] - truth[4:]).max()))
                v["vs_published"] = v["published_ms"] / v["ms_median"]
            pw["self_gauss_newton"] = {"published_ms": dit.PUBLISHED_MS["self_gauss_newton"], "ms_median": None,
                                       "reason": "not a ceres::Solve call: the reference's own Eigen loop (solver.hpp:387-462), nothing for a drop-in to replace"}
            pw["note"] = ("wall ms per call, construction of the ceres::Problem + Solve as the reference's timer spans them (solver.hpp:253-288), median of 199 "
                          "calls after the first (which pays HIP start-up); published = st17-ceres/img/release.png on an unstated desktop CPU. 6 unknowns / 40 "
                          "residuals: every LM iteration is the user's Evaluate on the host + ONE kernel launch and a poll of mapped memory "
                          "(small_dense.hip); the sized variant takes 9 iterations with the reference's own rotation Jacobian (8 published)")
            res["published_workload"] = pw
            d, cams = dit.run_ba(exe, s, tmp, reps=5, max_iterations=50, threads=1)
            # the C ABI underneath on the same scene: engine creation + solve to convergence + read-back
            t0 = time.perf_counter()
            e2 = st.BAEngine(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
            t1 = time.perf_counter()
            sg, _ = e2.solve()
            t2 = time.perf_counter()
            cams2, _ = e2.get_params()
            t3 = time.perf_counter()
            d["c_abi_seconds"] = {"create": t1 - t0, "solve": t2 - t1, "read_back": t3 - t2, "total": t3 - t0, "iterations": int(sg.num_iterations)}
            dq, dtp = pose_err(cams, cams2)
            d["pose_difference_to_c_abi"] = max(dq, dtp)
            d["solve_over_c_abi"] = d["seconds"]["solve"] / (t3 - t0)
            d["solve_over_device_solve"] = d["seconds"]["solve"] / max(d["seconds"]["device_solve"], 1e-12)
            gate = out.get("matched_result_gate")
            if gate and gate.get("passed"):
                # the CPU port's whole solve (same start, same iterations, matched result) against the same solve THROUGH THE API
                d["speedup_vs_cpu_port_through_the_api"] = gate["cpu_solve_seconds"] / d["seconds"]["solve"]
                d["speedup_note"] = (f"CPU port to convergence {gate['cpu_solve_seconds']:.2f} s ({gate['cpu_dense_solver']}) / ceres::Solve() "
                                     f"{d['seconds']['solve']:.3f} s wall (recognition, engine creation, device solve, write-back, end-point check)")
            d["note"] = ("seconds, median of 5: `build` = the caller's 10^6 AddResidualBlock (its cost with Ceres too), `solve` = ceres::Solve() wall, of "
                         "which recognise / pack / engine_create / device_solve / write_back / verify are the header's own phase timers; one host thread")
            res["drop_in"] = d
    except Exception as e:      # noqa: BLE001 -- a box without g++ must not lose the bench line
        res["drop_in"] = {"error": repr(e)}
    return res


def pg_bytes_per_edge(n_nodes, n_edges):
    ends = 2.0 * n_edges / max(n_nodes, 1)
    lin = 8 + 56 + 2 * 56 / ends + 48 + 2 * 288
    mv = 8 + 2 * 288 + 2 * 48 / ends + 96
    return lin, mv


def pmc_file(kind):
    """newest profiles/r<N>_pmc_<kind>.json (tools/pmc_round.sh) -> (relative path, dict) or (None, None)"""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_" + kind + ".json")):
        m = re.match(r"r(\d+)", os.path.basename(f))
        key = (int(m.group(1)), os.path.basename(f)) if m else None          # round number, then the tag's letter (r4_a < r4_e)
        if key and (best is None or key > best[0]):
            best = (key, f)
    if best is None:
        return None, None
    try:
        with open(best[1]) as fh:
            return os.path.relpath(best[1], ROOT), json.load(fh)
    except Exception:
        return None, None


def traffic_source(path, d, what):
    """says which file a `traffic` figure comes from, the build it was measured on and whether that is THIS build (same source hash)"""
    head = (d or {}).get("head", "unknown")
    mine = build_head()
    same = ("src:" in head and "src:" in mine and head.split("src:")[1] == mine.split("src:")[1])
    return {"file": path, "what": what, "measured_on_build": head, "this_build": mine, "same_sources_as_this_build": bool(same),
            "note": "rocprofv3 PMC passes are separate runs (tools/pmc_round.sh); the figure is recorded there, not measured inside bench.py"}


def build_head():
    """git head the library was built from (slam-tricks_amd/BUILD_HEAD, written by build.py where .git exists)"""
    try:
        return importlib.import_module("slam-tricks_amd.build").build_head()
    except Exception:
        return "unknown"


def bench_c4(args):
    """BASELINE config C4 (build-defined: the reference has no pose-graph code): 10 000 SE3 nodes, ~40 000 relative-pose
    edges, LM with inexact Newton steps: matrix-free PCG, two-level preconditioner (block Jacobi + rigid-body coarse space),
    forcing sequence (pg_engine.hip).  A "step" is one LM iteration (linearisation, coarse operator, PCG solve, trial point);
    `value` = LM iterations / s over whole solves to convergence from the same start, median of the repetitions.  N = 1 only
    (the sharded variant is covered by tests/test_sharding.py)."""
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libstba has no CPU fallback")
    torch.cuda.set_device(0)
    st = importlib.import_module("slam-tricks_amd")
    scenes = importlib.import_module("slam-tricks_amd.scenes")
    s = scenes.pose_graph_scene(n_nodes=args.pg_nodes, loops_per_node=3, seed=4)
    n, m = len(s["poses0"]), len(s["edge_i"])

    def fresh():
        return st.PGEngine(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
    for _ in range(max(1, args.warmup > 0)):
        fresh().solve(max_num_iterations=args.steps)
    reps = []
    for _ in range(max(1, args.reps)):
        e = fresh()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        summ, tr, pcg_total = e.solve(max_num_iterations=args.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        reps.append((dt, summ.num_iterations, pcg_total, summ.final_cost, summ.initial_cost, summ.termination_type, e.pcg_summary().as_dict(),
                     e.get_poses(), [int(x) for x in tr[:, 6]]))
    reps.sort(key=lambda r: r[0])
    dt, iters, pcg_total, fcost, icost, term, pcg_sum, gpu_poses, gpu_acc = reps[len(reps) // 2]
    # the same solve with EXACT steps (PCG to 1e-12): what the forcing sequence saves, and the run the oracle's trace is held against
    e = fresh()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    summ_x, tr_x, pcg_x = e.solve(max_num_iterations=args.steps, pcg=e.pcg_options(forcing_eta0=0.0, relative_tolerance=1e-12, max_iterations=2000))
    torch.cuda.synchronize()
    dt_x = time.perf_counter() - t0
    e = fresh()
    ms_lin, ms_mv = e.time_kernels(reps=200)
    b_lin, b_mv = pg_bytes_per_edge(n, m)
    # the dominant kernel of the production solve: the persistent PCG kernel, timed LIVE with hipEvents on the engine's stream around
    # every linear solve of one more whole solve (stba_lm_options::phase_timing -> stba_pcg_summary::linear_solve_ms)
    e = fresh()
    summ_t, _, pcg_t = e.solve(max_num_iterations=args.steps, phase_timing=1)
    ps_t = e.pcg_summary().as_dict()
    n_solves = max(int(ps_t["solves"]), 1)
    ms_pcg_launch = ps_t["linear_solve_ms"] / n_solves
    us_pcg_iter = 1e3 * ps_t["linear_solve_ms"] / max(pcg_t + n_solves, 1)      # (+1 per solve: the start-up pass has both exchanges too)
    # algorithmic bytes of one PCG iteration of the ASSEMBLED operator: one 6x6 block (288 B) + the remote node's u (48 B) per edge end,
    # diagonal block + M^-1 + P rows live in registers; u published (48 B per node) and nine partial sums all-gathered
    na = max(1, (n + 63) // 64)
    b_pcg_iter = (288.0 + 48.0) * 2 * m + 48.0 * n + 72.0 * na * na
    pcg_per_launch = (pcg_t + n_solves) / n_solves
    EXCHANGE_FLOOR_US = 6.0          # tools/exp/pcg_skeleton.hip, profiles/r5_pcg_skeleton.txt: both stamped exchanges of an iteration, nothing else
    products = pcg_total + iters                       # one more product per LM iteration (model decrease)
    out = {
        "metric": "LM iterations/sec, 10k-node / 40k-edge pose graph (BASELINE config 4)", "value": iters / dt, "unit": "LM iterations/s",
        "n_gpus": 1, "steps": int(iters), "warmup": int(args.warmup > 0), "ms_per_step": 1e3 * dt / max(iters, 1), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"C4 pose graph (build-defined): {n} SE3 nodes on the st4 sphere spiral, {m} relative-pose edges "
                               f"({n - 1} odometry + loop closures one revolution apart), LM to convergence, inexact steps: matrix-free PCG, "
                               "block Jacobi + rigid-body coarse space, Eisenstat-Walker forcing sequence",
                   "n_nodes": n, "n_edges": m, "parallelism": "single GPU"},
        "solve_seconds": dt, "lm_iterations": int(iters), "pcg_iterations": int(pcg_total), "pcg_iterations_per_lm_iteration": pcg_total / max(iters, 1),
        "pcg_hit_cap": int(pcg_sum["hit_cap"]), "pcg_summary": pcg_sum, "pcg_iterations_per_sec": pcg_total / dt,
        "edge_visits_per_sec": m * (products + 2.0 * iters) / dt,       # every product and every (trial / accepted) linearisation visits every edge
        "initial_cost": icost, "final_cost": fcost, "converged": bool(term == 0),
        "exact_steps": {"lm_iterations": int(summ_x.num_iterations), "pcg_iterations": int(pcg_x), "solve_seconds": dt_x,
                        "lm_iterations_per_sec": summ_x.num_iterations / dt_x, "final_cost": summ_x.final_cost,
                        "note": "the same solve with the PCG run to 1e-12 (forcing_eta0 = 0): the LM trace the oracle's is compared with"},
        "reps_solve_seconds": [r[0] for r in reps], "timing": "median of reps",
        "roofline": {"kernel": "pg_pcg_persistent_kernel (the whole PCG solve of an LM iteration: assembled operator, two stamped exchanges per iteration)",
                     "bound": "hbm", "achieved": b_pcg_iter * pcg_per_launch / (ms_pcg_launch * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": b_pcg_iter * pcg_per_launch / (ms_pcg_launch * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "ms_per_launch": ms_pcg_launch, "launches": n_solves, "pcg_iterations_per_launch": pcg_per_launch,
                     "algorithmic_bytes_per_pcg_iteration": b_pcg_iter, "algorithmic_bytes_per_launch": b_pcg_iter * pcg_per_launch,
                     "us_per_pcg_iteration": us_pcg_iter, "exchange_floor_us_per_pcg_iteration": EXCHANGE_FLOOR_US,
                     "frac_of_exchange_floor": EXCHANGE_FLOOR_US / max(us_pcg_iter, 1e-9),
                     "share_of_solve": ps_t["linear_solve_ms"] * 1e-3 / max(dt, 1e-12),
                     "note": f"{b_pcg_iter / 1e6:.1f} MB per PCG iteration, resident in the L2s after the first pass (8 x 4 MB): the HBM figure is the schema's, the "
                             "bound that binds is LATENCY -- an iteration is two dependent device-wide exchanges (neighbours' u; nine partial sums "
                             "all-gathered), 6.0 us for both in the bare skeleton; `frac_of_exchange_floor` is the fraction to read"},
        "roofline_edge_product": {"kernel": "pg_edge_product_kernel (the launch path's product, and the trial point's |J x|^2)",
                     "bound": "hbm", "achieved": b_mv * m / (ms_mv * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": b_mv * m / (ms_mv * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "ms_per_launch": ms_mv,
                     "algorithmic_bytes_per_edge": b_mv, "algorithmic_bytes_per_launch": b_mv * m},
        "roofline_linearize": {"kernel": "pg_linearize_kernel (residual + both 6x6 Jacobians per edge)", "bound": "hbm",
                               "achieved": b_lin * m / (ms_lin * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": b_lin * m / (ms_lin * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "ms_per_launch": ms_lin,
                               "algorithmic_bytes_per_edge": b_lin, "algorithmic_bytes_per_launch": b_lin * m},
        "edges_per_sec_linearize": m / (ms_lin * 1e-3), "edges_per_sec_matvec": m / (ms_mv * 1e-3),
        "build_head": build_head(),
    }
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_py as O       # cpu_baseline leg + matched-result gate only
        o = O.PG(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
        t0 = time.perf_counter()
        so, tro, cg, worst = o.solve_sparse(O.default_options(max_num_iterations=args.steps))
        tcpu = time.perf_counter() - t0
        dq = np.minimum(np.abs(gpu_poses[:, :4] - o.poses[:, :4]).max(1), np.abs(gpu_poses[:, :4] + o.poses[:, :4]).max(1)).max()
        dpose = float(max(dq, np.abs(gpu_poses[:, 4:] - o.poses[:, 4:]).max()))
        nx = min(len(tr_x), len(tro))
        gate = {"passed": bool(abs(fcost - so.final_cost) <= 1e-6 * so.final_cost and dpose <= 1e-5 and iters == so.num_iterations
                               and summ_x.num_iterations == so.num_iterations and np.allclose(tr_x[:nx, 0], tro[:nx, 0], rtol=1e-7)
                               and np.all(tr_x[:nx, 6] == tro[:nx, 6])),
                "final_cost_rel_diff": abs(fcost - so.final_cost) / so.final_cost, "pose_max_abs_diff": dpose,
                "lm_iterations_gpu": int(iters), "lm_iterations_gpu_exact_steps": int(summ_x.num_iterations), "lm_iterations_cpu": int(so.num_iterations),
                "exact_step_cost_trace_max_rel_diff": float(np.max(np.abs(tr_x[:nx, 0] - tro[:nx, 0]) / tro[:nx, 0])),
                "ate_gpu": float(O.pg_ate(s["poses_true"], gpu_poses)), "ate_cpu": float(O.pg_ate(s["poses_true"], o.poses)),
                "ate_initial": float(O.pg_ate(s["poses_true"], s["poses0"])),
                "tolerances": {"cost_rel": 1e-6, "pose": 1e-5, "exact_step_trace_rel": 1e-7}}
        out["matched_result_gate"] = gate
        out["cpu_baseline"] = {"value": so.num_iterations / tcpu, "unit": "LM iterations/s", "cores": 16, "kind": "port", "cpu_model": cpu_model(),
                               "sample": f"one whole solve of the same C4 problem to convergence ({so.num_iterations} LM iterations, {cg} CG iterations) by "
                                         "oracle/liboracle.so: orc_pg_solve_sparse -- the same LM with EXACT steps, the normal equations solved matrix-free by "
                                         f"conjugate gradients to 1e-13 and certified (worst explicit residual {worst:.1e}), OpenMP teams of at most 16 threads, "
                                         f"{tcpu:.1f} s wall", "seconds": tcpu, "cg_iterations": int(cg)}
        if gate["passed"]:
            out["speedup_vs_cpu_port"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the K timed steps; the median is reported")
    ap.add_argument("--cams", type=int, default=1000)
    ap.add_argument("--pts", type=int, default=100000)
    ap.add_argument("--obs-per-pt", type=int, default=10)
    ap.add_argument("--dense-visibility", action="store_true",
                    help="a scene where every camera sees every landmark in front of it (143 degree field of view, no observation cap): the Schur "
                         "complement then runs as one symmetric product on the matrix cores (DESIGN.md 4); with few cameras and many landmarks "
                         "(--cams 100 --pts 200000) this is the workload landmark sharding scales on")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-library-baseline", action="store_true", help="skip the rocSOLVER potrf cross-check")
    ap.add_argument("--no-drop-in", action="store_true",
                    help="skip the wall-clock of the drop-in path (ceres::Solve through include/stba/ceres.h at C5, the reference's published PnP workload)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: CPU rendezvous test of the launch path")
    ap.add_argument("--hook", default="native", choices=["native", "torch"], help="cross-rank sum: native RCCL communicator | torch.distributed hook")
    ap.add_argument("--config", default="c5", choices=["c5", "c4"], help="c5: the headline BA workload (default); c4: the 10k-node pose graph")
    ap.add_argument("--pg-nodes", type=int, default=10000)
    ap.add_argument("--second-scene", action="store_true",
                    help="also time the LANDMARK-HEAVY scene (--second-cams x --second-pts, 10 observations per landmark) and emit it under the key "
                         "`landmark_heavy` with its own predicted_scaling; on by itself with --gpus N > 1: a driver with several GPUs then records the "
                         "headline C5 curve (flat: the replicated factorisation) and the one landmark sharding can scale on, in one call")
    ap.add_argument("--no-second-scene", action="store_true")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal of the N-rank path on a one-GPU box: all ranks on device 0 (use with --backend gloo --hook torch; the times mean nothing)")
    ap.add_argument("--second-cams", type=int, default=100)
    ap.add_argument("--second-pts", type=int, default=1000000)
    args = ap.parse_args()
    want_second = (args.second_scene or args.gpus > 1) and not args.no_second_scene and not args.dense_visibility
    if args.config == "c4":
        if args.gpus != 1:
            raise SystemExit("bench.py --config c4 measures one GPU")
        if args.steps == 100:
            args.steps = 50          # (LM iteration limit of the solve: Ceres' default)
        return bench_c4(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch(args))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_gpu:
        local_rank = 0          # (the N-rank path rehearsed on a box with ONE device: every rank on device 0, gloo + the torch hook)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python bench.py --gpus N does it itself)")
    import torch
    have_gpu = torch.cuda.is_available()
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if have_gpu:
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=args.backend if have_gpu else "gloo", rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus
    elif have_gpu:
        torch.cuda.set_device(0)

    st = importlib.import_module("slam-tricks_amd")
    sharding = importlib.import_module("slam-tricks_amd.sharding")
    s = load_scene(args, rank)
    n_cams, n_pts, n_obs = len(s["cams0"]), len(s["pts0"]), len(s["obs_cam"])
    sh = sharding.make_shard(s, rank, world)

    if st.device_count() <= 0:
        if world > 1 and args.backend == "gloo":
            # launch-path check on a box without GPUs: rendezvous, sharding, the max-over-ranks reduction and the
            # JSON line are exercised; the product has no CPU fallback, so there is nothing to time
            counts = torch.zeros(world, dtype=torch.float64)
            counts[rank] = float(len(sh["obs_cam"]))
            dist.all_reduce(counts)
            second = None
            if want_second:
                s2 = load_second_scene(args, rank)
                sh2 = sharding.make_shard(s2, rank, world)
                c2 = torch.zeros(world, dtype=torch.float64)
                c2[rank] = float(len(sh2["obs_cam"]))
                dist.all_reduce(c2)
                second = {"value": None, "skipped": "no HIP device", "n_obs": int(len(s2["obs_cam"])), "n_cams": int(len(s2["cams0"])),
                          "n_pts": int(len(s2["pts0"])), "observations_per_rank": [int(v) for v in c2.tolist()],
                          "predicted_scaling": None, "camera_blocks_identical_on_all_ranks": None}
            dist.barrier()
            if rank == 0:
                # (the keys that explain an N > 1 line are present, without values: nothing ran)
                print(json.dumps({"metric": "LM iterations/sec + residuals/sec, 1k-cam/100k-pt BA", "value": None,
                                  "unit": "LM iterations/s", "n_gpus": world, "ranks": dist.get_world_size(),
                                  "skipped": "no HIP device: libstba has no CPU fallback",
                                  "sharded_observations": int(counts.sum().item()), "n_obs": n_obs,
                                  "observations_per_rank": [int(v) for v in counts.tolist()],
                                  "phase_ms_per_step": None, "allreduce_ms": None, "allreduce_bytes": None,
                                  "allreduce_calls_per_step": None, "camera_blocks_identical_on_all_ranks": None,
                                  "landmark_heavy": second}), flush=True)
            dist.destroy_process_group()
            return
        raise SystemExit("bench.py needs an MI355X: libstba has no CPU fallback")

    # the engine runs on its own (non-default) stream; collectives are enqueued on the same stream
    stream_obj = torch.cuda.Stream()
    stream = stream_obj.cuda_stream
    eng, comm, collective = make_engine(st, sharding, torch, dist, sh, stream, rank, world, local_rank, have_gpu, args)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- warm-up, then R repetitions of exactly K timed LM iterations from the same start point
    if args.warmup > 0:
        eng.lm_iterations(args.warmup)
    rep_ms, summ = [], None
    for _ in range(max(1, args.reps)):
        eng.set_params(sh["cams0"], sh["pts0"])
        sync()
        t0 = time.perf_counter()
        summ, trace = eng.lm_iterations(args.steps)
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        rep_ms.append(1e3 * dt / args.steps)
    ms_step = float(np.median(rep_ms))
    it_per_s = 1e3 / ms_step
    cams_same, _ = ranks_hold_identical_cameras(eng, torch, dist, world, have_gpu, args)
    rank_check = LAST_RANK_CHECK
    out = {
        "metric": "LM iterations/sec + residuals/sec, 1k-cam/100k-pt BA",
        "value": it_per_s, "unit": "LM iterations/s",
        "residuals_per_sec": it_per_s * 2.0 * n_obs,
        "n_gpus": world, "ranks": world if dist is None else dist.get_world_size(), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "reps": len(rep_ms), "reps_ms_per_step": rep_ms, "timing": "median of reps, max over ranks",
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("dense-visibility synthetic BA (not a BASELINE config; the matrix-core form of the Schur complement): "
                                if args.dense_visibility else "C5 large synthetic BA: ") + f"{n_cams} cams, {n_pts} pts, {n_obs} obs "
                               f"({2 * n_obs} residuals), Schur + dense {6 * n_cams}x{6 * n_cams} Cholesky, "
                               "st20 spiral/cube scene seed 20, pixel noise 1e-3",
                   "parallelism": f"landmark-shard x{world}" if world > 1 else "single GPU",
                   "collective": collective, "n_cams": n_cams, "n_pts": n_pts, "n_obs": n_obs},
        "final_cost": summ.final_cost,
        "camera_blocks_identical_on_all_ranks": bool(cams_same),    # every rank factors the same reduced system: compared in-run, bit for bit
        "rank_check": rank_check,       # (several ranks: largest distance of a rank's camera blocks from rank 0's, factorisation time-outs per rank)
    }
    # Device time per phase: ONE MORE run of K steps with stba_lm_options::phase_timing on (hipEvents between the phases; the
    # timed repetitions above run without them: an event record costs ~5 us of idle GPU and an iteration would take eight),
    # per step; with several ranks the MAXIMUM over ranks, plus what explains an N > 1 line: the cross-rank sum's device time
    # and bytes per step and the observations each rank holds (the Cholesky of the reduced system is replicated on every
    # rank: DESIGN.md 6)
    eng.set_params(sh["cams0"], sh["pts0"])
    sync()
    t0 = time.perf_counter()
    summ, _ = eng.lm_iterations(args.steps, phase_timing=1)
    sync()
    out["instrumented_ms_per_step"] = 1e3 * (time.perf_counter() - t0) / args.steps
    phase_keys = ("ms_linearize", "ms_schur", "ms_solve", "ms_backsub", "ms_cost", "ms_allreduce")
    phases = np.array([getattr(summ, k) / args.steps for k in phase_keys], dtype=np.float64)
    local_counts = np.zeros(world, dtype=np.float64)
    local_counts[rank] = float(len(sh["obs_cam"]))
    if world > 1:
        dev = "cuda" if (have_gpu and args.backend == "nccl") else "cpu"
        tp = torch.tensor(phases, dtype=torch.float64, device=dev)
        dist.all_reduce(tp, op=dist.ReduceOp.MAX)
        phases = tp.cpu().numpy()
        tc = torch.tensor(local_counts, dtype=torch.float64, device=dev)
        dist.all_reduce(tc)
        local_counts = tc.cpu().numpy()
    out["phase_ms_per_step"] = {k: float(v) for k, v in zip(phase_keys[:5], phases[:5])}
    out["phase_ms_per_step"]["timing"] = "hipEvents on the engine stream, a separate instrumented run (instrumented_ms_per_step)" + (", max over ranks" if world > 1 else "")
    out["allreduce_ms"] = float(phases[5])                      # per step, inside ms_schur
    out["allreduce_bytes"] = float(summ.allreduce_bytes / max(1, summ.allreduce_calls))      # per call and rank
    out["allreduce_calls_per_step"] = float(summ.allreduce_calls / args.steps)
    out["observations_per_rank"] = [int(v) for v in local_counts]

    if want_second:
        # the landmark-heavy scene, same protocol, all ranks (fewer steps: an iteration is ~10x the observations of C5)
        s2 = load_second_scene(args, rank)
        out["landmark_heavy"] = time_scene(st, sharding, torch, dist, s2, args, rank, world, local_rank, have_gpu,
                                           steps=max(2, min(args.steps, 20)), reps=min(args.reps, 3), warmup=min(args.warmup, 2))
        out["landmark_heavy"]["note"] = ("not a BASELINE config: the workload landmark sharding scales on (DESIGN.md 6), timed next to the headline so "
                                         "that one multi-GPU call records both curves")
    if rank == 0:
        # ---- roofline legs, measured live with hipEvents on the engine's stream
        ms_jac = eng.time_linearize(20)
        local_obs = len(sh["obs_cam"])
        jac_bytes = BYTES_PER_OBS_JAC * local_obs
        jac_gbs = jac_bytes / (ms_jac * 1e-3) / 1e9
        nred = 6 * n_cams
        ms_factor, ms_bwd = st.cholesky_time_split(nred, reps=15)   # medians; ms_factor = the persistent kernel alone (events right around it)
        chol_flops = nred ** 3 / 3.0 + nred ** 2 / 2.0            # algorithmic flops of one n x n Cholesky
        chol_tflops = chol_flops / (ms_factor * 1e-3) / 1e12
        prof = st.cholesky_profile(nred)                          # stage-per-kernel schedule (diagnostic)
        syrk_tflops = prof["syrk_flops"] / (prof["ms_syrk"] * 1e-3) / 1e12 if prof["ms_syrk"] > 0 else 0.0
        # HBM bytes per launch from the PMC passes of the round (tools/pmc_round.sh -> profiles/r<N>_pmc_*.json, stamped with the
        # build they were taken on); only valid for the C5 shape they were collected on
        pj_path, pj = pmc_file("jacobian")
        traffic = pj["hbm_bytes_per_launch"] if (pj and local_obs == 1000000) else None
        # IN the LM iteration the kernel runs behind the back-substitution with cold caches: its time there is the 'cost' phase of the
        # instrumented run (the speculative linearisation at the trial point IS that kernel; it includes the event packets around it);
        # the back-to-back figure (the kernel alone, 20 launches) is kept beside it
        ms_jac_iter = float(phases[4]) if (world == 1 and phases[4] > 0) else None
        ms_for_roof = ms_jac_iter if ms_jac_iter else ms_jac
        roof_jac = {"kernel": "ba_linearize_kernel<cams-in-LDS, with-Jacobian> (residual + compact 64 B Jacobian per observation)", "bound": "hbm",
                    "achieved": jac_bytes / (ms_for_roof * 1e-3) / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": jac_bytes / (ms_for_roof * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_source": traffic_source(pj_path, pj, "FETCH_SIZE (doubled, gfx950) + WRITE_SIZE per launch, C5 shape"),
                    "ms_per_launch": ms_for_roof,
                    "timing": "in the LM iteration: phase 'ms_cost' of the instrumented run (hipEvents around the kernel on the engine stream)" if ms_jac_iter
                              else "back-to-back launches of the kernel alone",
                    "solo": {"ms_per_launch": ms_jac, "achieved": jac_gbs, "frac": jac_gbs / HBM_PEAK_GBS,
                             "timing": "the kernel alone, 20 back-to-back launches (stba_ba_time_linearize)"},
                    "algorithmic_bytes_per_launch": jac_bytes, "algorithmic_bytes_per_observation": BYTES_PER_OBS_JAC}
        pc_path, pc = pmc_file("chol_mfma")
        pc6 = (pc or {}).get("sizes", {}).get("6000") if nred == 6000 else None
        chol_traffic = pc6.get("hbm_bytes_per_launch") if pc6 else None
        roof_chol = {"kernel": "chol_mega_kernel (persistent dataflow Cholesky, v_mfma_f64_16x16x4_f64)",
                     "bound": "mfma", "achieved": chol_tflops, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": chol_tflops / FP64_MFMA_PEAK_TFLOPS, "traffic": chol_traffic,
                     "traffic_source": traffic_source(pc_path, pc, "FETCH_SIZE (doubled, gfx950) + WRITE_SIZE per factorisation at n = 6000"),
                     "mfma_utilisation_pmc": pc6.get("mfma_utilisation") if pc6 else None,
                     "l2_hit_rate_pmc": pc6.get("l2_hit_rate") if pc6 else None,
                     "ms_per_launch": ms_factor,
                     "timing": "median of 15 factorisations of a synthetic SPD matrix of the C5 size, hipEvents on the engine's stream right "
                               "in front of and behind chol_mega_kernel (the quantity rocprofv3 --kernel-trace reports for it)",
                     "algorithmic_flops_per_launch": chol_flops, "launches_per_lm_iteration": 1,
                     "microbench_ceiling": FP64_MFMA_MEASURED_CEILING_TFLOPS,
                     "frac_of_microbench_ceiling": chol_tflops / FP64_MFMA_MEASURED_CEILING_TFLOPS}
        # the Schur-complement kernel.  Round 6 settled what bounds it: with the LDS-atomic collisions dealt away by the host it gains
        # 1-3 %, without atomics at all (register sums per block run) it LOSES 37 % on the landmark-heavy scene -- it is bound by what
        # it gathers (profiles/r6_schur_rot_rank.txt, r6_landmark_range_slices_register_runs.txt).
        # The LDS-atomic rate is kept as a second figure (peak = the microbenchmark's rate with THIS address pattern,
        # profiles/lds_atomic_f64_microbench.txt, 2.38 lane-ops per cycle and CU).
        ms_schur, schur_atomics, schur_pairs = eng.time_schur(10)
        pk_path, pk = pmc_file("assembly_kernels")
        schur_traffic = None
        if pk and local_obs == 1000000:
            for kname, kd in pk.get("kernels", {}).items():
                if "schur_pairs" in kname and "hbm_bytes_per_launch_raw" in kd:
                    schur_traffic = kd["hbm_bytes_per_launch_raw"]
        LDS_ATOMIC_PEAK = 1459.2      # G lane-ops/s, all 256 CUs, Schur pattern (7.5 per cycle and CU without conflicts: 4617 G/s)
        lda_s = ((nred + 1 + 127) // 128) * 128
        zero_bytes = sum(6 * min(lda_s, ((6 * c + 5) // 128 + 1) * 128) * 8 for c in range(n_cams))
        # ALGORITHMIC bytes = everything the step must touch, once: the observations' 64 B Jacobian records, the landmarks' 48 B inverse
        # blocks, the 16 B plan record of every pair, the lower triangle of S written once (zeros included).  What the pair formulation
        # GATHERS is several times that (every pair asks for two records and a block again: 192 B per pair) -- reported beside it.
        n_local_pts_s = len(sh["pts0"])
        schur_bytes = 64.0 * local_obs + 48.0 * n_local_pts_s + 16.0 * schur_pairs + zero_bytes
        gathered_bytes = 192.0 * schur_pairs + zero_bytes
        schur_gbs = schur_bytes / (ms_schur * 1e-3) / 1e9
        out["roofline_schur"] = {"kernel": "ba_schur_pairs_kernel (row-wise Schur complement, LDS accumulation; camera blocks on the way)",
                                 "bound": "hbm", "achieved": schur_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": schur_gbs / HBM_PEAK_GBS,
                                 "algorithmic_bytes_per_launch": schur_bytes, "zero_fill_bytes": zero_bytes,
                                 "gathered": {"bytes_per_launch": gathered_bytes, "bytes_per_pair": 192, "GB/s": gathered_bytes / (ms_schur * 1e-3) / 1e9,
                                              "frac": gathered_bytes / (ms_schur * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                              "what": "what the pair loop asks for (two 64 B records, a 48 B block and a 16 B plan record per pair) + the zero fill: "
                                                      "the caches serve part of it, the counter traffic is what reached the fabric"},
                                 "ms_per_launch": ms_schur, "pairs_per_launch": schur_pairs,
                                 "lds_atomic": {"achieved": schur_atomics / (ms_schur * 1e-3) / 1e9, "peak": LDS_ATOMIC_PEAK, "unit": "G ds_add_f64 lane-ops/s",
                                                "frac": schur_atomics / (ms_schur * 1e-3) / 1e9 / LDS_ATOMIC_PEAK, "per_launch": schur_atomics,
                                                "floor_ms": schur_atomics / (LDS_ATOMIC_PEAK * 1e9) * 1e3},
                                 "traffic": schur_traffic,
                                 "traffic_source": traffic_source(pk_path, pk, "FETCH_SIZE + WRITE_SIZE per launch at C5, RAW (the pair loop gathers 64-B records: "
                                                                  "narrow requests, not the wide coalesced reads the gfx950 doubling is for)")}
        if eng.schur_mode() == eng.SCHUR_DENSE:
            # dense visibility: S = -(Y Y^T) on the matrix cores; algorithmic flops of the lower triangle: n^2 / 2 entries x K x 2
            n_local_pts = len(sh["pts0"])
            yyt_flops = float(nred) * nred * 3.0 * n_local_pts
            out["roofline_schur"] = {"kernel": "ba_schur_yfill_kernel + yyt_tile_kernel (dense visibility: S = -(Y Y^T), v_mfma_f64_16x16x4_f64; timing: the whole "
                                               "Schur step, Y fill, product, right-hand side and camera blocks)",
                                     "bound": "mfma", "achieved": yyt_flops / (ms_schur * 1e-3) / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                     "frac": yyt_flops / (ms_schur * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, "ms_per_launch": ms_schur,
                                     "algorithmic_flops_per_launch": yyt_flops, "traffic": None}
        # the dominant kernel by device time carries the headline roofline object
        out["roofline"] = roof_chol if ms_factor > ms_jac else roof_jac
        if eng.schur_mode() == eng.SCHUR_DENSE and ms_schur > max(ms_factor, ms_jac): out["roofline"] = out["roofline_schur"]
        out["roofline_jacobian"] = roof_jac
        out["roofline_mfma"] = roof_chol
        out["cholesky_ms"] = {"factor_persistent_kernel": ms_factor, "backward": ms_bwd,
                              "stage_kernels_serial": {"diag": prof["ms_diag"], "trsm": prof["ms_trsm"],
                                                       "syrk": prof["ms_syrk"], "backward": prof["ms_bwd"],
                                                       "syrk_tflops": syrk_tflops}}

        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle_py as O       # cpu_baseline leg only: the oracle is the thing timed / the checker here
            ncpu = os.cpu_count() or 1
            model = cpu_model()
            quota, affinity = cpu_quota()
            usable = int(max(1, min(affinity, round(quota) if quota else affinity)))
            out["host_cpu"] = {"model": model, "logical_cores": ncpu, "affinity_cores": affinity, "cgroup_cpu_quota": quota, "usable_cores": usable,
                               "note": "the CPU legs run on `usable_cores` = min(affinity, cgroup quota): thread teams larger than the quota are throttled by "
                                       "the container runtime and run slower (cpu_thread_sweep shows it)"}

            def fresh():
                return O.BA(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])

            # ---- matched-result gate: same start, both legs run to CONVERGENCE (Ceres' default stopping rules,
            # test_ceres.h:148; north_star: "same converged parameters"); the oracle factors with LAPACK here so that the
            # ~30 iterations take seconds
            ob = fresh()
            gate_threads = min(usable, 64)
            gate_blas = O.use_lapack(True, threads=gate_threads)
            tg = time.perf_counter()
            try:
                so, tro = ob.solve(num_threads=gate_threads)
            finally:
                O.use_lapack(False)
            t_gate = time.perf_counter() - tg
            eng.set_params(s["cams0"], s["pts0"])
            sg, trg = eng.solve()
            n_gate = int(so.num_iterations)
            cams_g, _ = eng.get_params()
            rel_cost = abs(sg.final_cost - so.final_cost) / max(abs(so.final_cost), 1e-300)
            dq, dtp = pose_err(cams_g, ob.cams)
            same_seq = bool(sg.num_iterations == n_gate and np.array_equal(trg[: n_gate + 1, 6], tro[: n_gate + 1, 6]))
            matched = bool(rel_cost <= 1e-6 and dq <= 1e-5 and dtp <= 1e-5 and same_seq and sg.termination_type == 0 and so.termination_type == 0)
            out["matched_result_gate"] = {"run": "to convergence on both legs", "iterations_gpu": int(sg.num_iterations), "iterations_cpu": n_gate,
                                          "gpu_final_cost": sg.final_cost, "cpu_final_cost": so.final_cost,
                                          "relative_cost_difference": rel_cost, "pose_dq": dq, "pose_dt": dtp,
                                          "same_accept_reject_sequence": same_seq, "gpu_solve_seconds": sg.seconds_total,
                                          "cpu_solve_seconds": t_gate, "cpu_threads": gate_threads, "cpu_dense_solver": gate_blas or "oracle C Cholesky",
                                          "tolerances": {"cost_rel": 1e-6, "pose": 1e-5}, "passed": matched}

            def time_oracle(threads, budget_s, runs):
                """median over `runs` timed runs of n fixed-work LM iterations of the SAME C5 problem on the host;
                n is calibrated from one iteration so that all runs together fit the budget (bounded sample)"""
                ba = fresh()
                tc = time.perf_counter()
                ba.solve(fixed_iterations=1, num_threads=threads)
                t1 = time.perf_counter() - tc
                n = int(min(10, max(1, round(budget_s / runs / max(t1, 1e-3)))))
                rates, total = [], t1
                for _ in range(runs):
                    ba = fresh()
                    tc = time.perf_counter()
                    ba.solve(fixed_iterations=n, num_threads=threads)
                    d = time.perf_counter() - tc
                    total += d
                    rates.append(n / d)
                return float(np.median(rates)), n, runs, total
            # the reference pins num_threads = 1 (test_ceres.h:143); also report the best OpenMP setting
            # (measured on the GPU box's 2 x EPYC 9575F: 16 threads is the optimum of the oracle)
            it1, n1, r1, d1 = time_oracle(1, 8.0, 1)
            # thread-count sweep ON THIS BOX (every speed-up used to be quoted against 16 of the box's logical cores
            # without a look at the others): two fixed-work iterations per setting, the best one is then measured properly
            sweep = {}
            for th in sorted({max(1, usable // 2), usable, min(ncpu, 2 * usable), min(ncpu, 4 * usable)}):
                ba = fresh()
                ba.solve(fixed_iterations=1, num_threads=th)            # (first touch / thread team start-up outside the timing)
                ba = fresh()
                tc = time.perf_counter()
                ba.solve(fixed_iterations=2, num_threads=th)
                sweep[th] = 2.0 / (time.perf_counter() - tc)
            thb = max(sweep, key=sweep.get)
            out["cpu_thread_sweep"] = {"lm_iterations_per_sec_by_threads": {str(k): v for k, v in sweep.items()}, "best": thb,
                                       "logical_cores_on_box": ncpu, "usable_cores": usable, "cgroup_cpu_quota": quota,
                                       "sample": "2 fixed-work LM iterations of the C5 problem per setting (half, once, twice and four times the usable "
                                                 "cores), oracle C Cholesky; round 6: the port's assembly no longer walks every observation in every thread "
                                                 "(oracle.c, ba_index) -- what is left above `usable_cores` is the container's CPU quota, not the port"}
            itb, nb_, rb, db = time_oracle(thb, 8.0, 5)
            plain = {"value": itb, "unit": "LM iterations/s", "cores": thb, "kind": "port", "cpu_model": model,
                     "logical_cores_on_box": ncpu, "residuals_per_sec": itb * 2.0 * n_obs,
                     "sample": f"median of {rb} runs of {nb_} fixed-work LM iterations of the same C5 problem, "
                               f"oracle/liboracle.so (C port: OpenMP x{thb}, dense blocked Cholesky), {db:.1f} s wall",
                     "seconds": db}
            out["cpu_baseline"] = plain
            # the same port with the reduced system factored by LAPACK (the OpenBLAS scipy ships): the fair opponent for a
            # dense 6000 x 6000 FP64 Cholesky.  The faster of the two is THE cpu_baseline.
            blas_threads = min(usable, 64)    # (the GPU boxes' quota is 16 CPUs: dpotrf n = 6000 runs 80 ms at 16 threads, 258 ms at 64 -- throttled; tools/cpu_blas_check.py)
            blas = O.use_lapack(True, threads=blas_threads)
            if blas:
                try:
                    itl, nl, rl, dl = time_oracle(thb, 8.0, 5)
                    lap = {"value": itl, "unit": "LM iterations/s", "cores": max(thb, blas_threads), "kind": "port", "cpu_model": model,
                           "logical_cores_on_box": ncpu, "residuals_per_sec": itl * 2.0 * n_obs,
                           "sample": f"median of {rl} runs of {nl} fixed-work LM iterations of the same C5 problem, oracle/liboracle.so "
                                     f"(C port: OpenMP x{thb}) with the reduced system factored by LAPACK dpotrf/dpotrs, {blas}, {dl:.1f} s wall",
                           "seconds": dl}
                    if itl > itb:
                        out["cpu_baseline"] = lap
                        out["cpu_baseline_plain_c_cholesky"] = plain
                        itb = itl
                    else:
                        out["cpu_baseline_lapack_cholesky"] = lap
                finally:
                    O.use_lapack(False)
            out["cpu_baseline_single_thread"] = {"value": it1, "unit": "LM iterations/s", "cores": 1, "kind": "port", "cpu_model": model,
                                                 "sample": f"{r1} run of {n1} iterations, num_threads = 1 as the reference pins "
                                                           f"(test_ceres.h:143), {d1:.1f} s wall", "seconds": d1}
            if matched:      # a speed-up is only printed for matched results
                out["speedup_vs_cpu_port_single_thread"] = it_per_s / it1
                out["speedup_vs_cpu_port"] = it_per_s / itb
            else:
                out["speedup_vs_cpu_port"] = None
                out["speedup_note"] = "withheld: the GPU leg and the CPU leg did not match (see matched_result_gate)"
            # opportunistic real-Ceres row
            scene_file = None
            try:
                if os.path.exists("/usr/include/ceres/ceres.h") or os.path.exists("/usr/local/include/ceres/ceres.h"):
                    import struct
                    scene_file = os.path.join(os.environ.get("TMPDIR", "/tmp"), "stba_c5_scene.bin")
                    with open(scene_file, "wb") as f:
                        f.write(struct.pack("iii", n_cams, n_pts, n_obs))
                        for k, ty in (("cams0", np.float64), ("pts0", np.float64), ("obs_cam", np.int32), ("obs_pt", np.int32), ("obs_feat", np.float64)):
                            f.write(np.ascontiguousarray(s[k], ty).tobytes())
                        f.write(np.ascontiguousarray(s["cam_fixed"][:, 0], np.uint8).tobytes())
            except Exception:
                scene_file = None
            out["ceres_baseline"] = ceres_row(scene_file, n_gate) if scene_file else ceres_row("", n_gate)
        if world == 1 and not args.no_drop_in and not args.dense_visibility:
            out.update(drop_in_rows(s, eng, out))
        out["build_head"] = build_head()
        # ---- what landmark sharding can and cannot buy on this problem, from THIS run's phase times (a model, printed so that a
        # measured N-GPU point can be held against it): the factorisation + backward substitution are replicated on every rank,
        # the Jacobian / Schur / back-substitution / trial-cost work divides by N, and one packed all-reduce of the reduced system
        # per build is added (bounds: a ring over one xGMI link, 48 GB/s each way; reduce-scatter + all-gather over all N - 1 links)
        if local_counts.sum() > 0:
            ar_bytes = out["allreduce_bytes"] if world > 1 else (4.0 * nred * (nred + 1) if args.dense_visibility else 40.7e6 * (n_cams / 1000.0) ** 2)
            out["predicted_scaling"] = predicted_scaling(out["phase_ms_per_step"], ms_step, out["allreduce_ms"], ar_bytes, world,
                                                         "strong scaling of C5 is bounded by the replicated factorisation (DESIGN.md 6); a problem with few "
                                                         "cameras and many landmarks is where landmark sharding scales: see the key `landmark_heavy` "
                                                         "(emitted with --gpus N > 1, or --second-scene)")
        # ---- library cross-check: rocSOLVER's potrf + potrs of the same size on the same box, a stated baseline
        # like the CPU leg.  librocsolver is dlopen'ed by the TOOL, never by libstba.
        if world == 1 and not args.no_library_baseline:
            try:
                import importlib.util as ilu
                spec = ilu.spec_from_file_location("rocsolver_potrf", os.path.join(ROOT, "tools", "rocsolver_potrf.py"))
                mod = ilu.module_from_spec(spec)
                spec.loader.exec_module(mod)
                lb = mod.time_potrf(nred, reps=3)
                if lb.get("found"):
                    out["library_baseline"] = {"library": "rocSOLVER (rocsolver_dpotrf + rocsolver_dpotrs, one right-hand side), dlopen'ed by tools/rocsolver_potrf.py",
                                               "n": nred, "potrf_ms": lb["potrf_ms_median"], "potrs_ms": lb["potrs_ms_median"], "potrf_tflops": lb["potrf_tflops"],
                                               "stba_factor_ms": ms_factor, "stba_backward_ms": ms_bwd,
                                               "rocsolver_over_stba": (lb["potrf_ms_median"] + lb["potrs_ms_median"]) / (ms_factor + ms_bwd),
                                               "kind": "baseline only: never linked into or called by libstba.so"}
                else:
                    out["library_baseline"] = {"found": False}
            except Exception as e:      # noqa: BLE001
                out["library_baseline"] = {"found": False, "error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if comm is not None:
        eng.close()
        comm.close()


if __name__ == "__main__":
    main()
