#!/usr/bin/env python
"""bench.py -- LM iterations/s + residuals/s on the 1k-camera / 100k-point synthetic BA (BASELINE
config C5: 1 000 cameras, 100 000 landmarks, 1 000 000 reprojection observations, FP64).

A "step" is ONE Levenberg-Marquardt iteration of the hot path over the whole problem:
residual+Jacobian kernel -> block J^T J / J^T r -> landmark Schur complement -> dense FP64
Cholesky of the 6000 x 6000 reduced camera system (MFMA) -> back-substitution -> manifold
update -> residual kernel at the trial point (stba_ba_lm_iterations: every iteration does all
of that, accepted or not).  Inputs are resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--cams C --pts P --obs-per-pt M]

N > 1: one process per GPU (torch.distributed.run, backend nccl = RCCL); landmarks are sharded
across ranks, the cameras are replicated, and one RCCL all-reduce per build carries the reduced
camera system (strong scaling: the problem is fixed).  Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix peak (SURVEY.md 8d / AMD datasheet)
FP64_MFMA_MEASURED_CEILING_TFLOPS = 77.3   # profiles/mfma_f64_microbench.txt (pure-MFMA loop, 512-thread blocks, this chip)
BYTES_PER_OBS_JAC = 186.5    # materialised residual+Jacobian kernel, 10 obs/landmark


def load_scene(args, rank):
    scenes = importlib.import_module("slam-tricks_amd.scenes")
    tag = f"c{args.cams}_p{args.pts}_m{args.obs_per_pt}_s20"
    cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"stba_scene_{tag}.npz")
    if os.path.exists(cache):
        try:
            z = np.load(cache)
            return {k: z[k] for k in z.files}
        except Exception:
            pass
    s = scenes.st20_scene(n_cams=args.cams, n_pts=args.pts, max_obs_per_pt=args.obs_per_pt, seed=20,
                          pix_noise=1e-3)
    if rank == 0:
        try:
            np.savez(cache + f".tmp{os.getpid()}.npz", **s)
            os.replace(cache + f".tmp{os.getpid()}.npz", cache)
        except Exception:
            pass
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cams", type=int, default=1000)
    ap.add_argument("--pts", type=int, default=100000)
    ap.add_argument("--obs-per-pt", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=0, help="0 = calibrate to ~15 s")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(0)

    st = importlib.import_module("slam-tricks_amd")
    if st.device_count() <= 0:
        raise SystemExit("bench.py needs an MI355X: libstba has no CPU fallback")

    s = load_scene(args, rank)
    n_cams, n_pts, n_obs = len(s["cams0"]), len(s["pts0"]), len(s["obs_cam"])
    sharding = importlib.import_module("slam-tricks_amd.sharding")
    sh = sharding.make_shard(s, rank, world)
    stream = torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else None
    eng = st.BAEngine(sh["cams0"], sh["pts0"], sh["obs_cam"], sh["obs_pt"], sh["obs_feat"], sh["cam_fixed"],
                      stream=stream)
    if world > 1:
        eng.set_allreduce(sharding.torch_allreduce_hook(dist, torch), rank, world)

    def sync():
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- warm-up, then exactly K timed LM iterations
    if args.warmup > 0:
        eng.lm_iterations(args.warmup)
    sync()
    t0 = time.perf_counter()
    summ, trace = eng.lm_iterations(args.steps)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    it_per_s = args.steps / dt
    out = {
        "metric": "LM iterations/sec + residuals/sec, 1k-cam/100k-pt BA",
        "value": it_per_s, "unit": "LM iterations/s",
        "residuals_per_sec": it_per_s * 2.0 * n_obs,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"C5 large synthetic BA: {n_cams} cams, {n_pts} pts, {n_obs} obs "
                               f"({2 * n_obs} residuals), Schur + dense {6 * n_cams}x{6 * n_cams} Cholesky, "
                               "st20 spiral/cube scene seed 20, pixel noise 1e-3",
                   "parallelism": f"landmark-shard x{world}" if world > 1 else "single GPU",
                   "n_cams": n_cams, "n_pts": n_pts, "n_obs": n_obs},
        "phase_ms_per_step": {k: getattr(summ, k) / args.steps for k in
                              ("ms_linearize", "ms_schur", "ms_solve", "ms_backsub", "ms_cost")},
        "final_cost": summ.final_cost,
    }

    if rank == 0:
        # ---- roofline legs, measured live with hipEvents on the engine's stream
        ms_jac = eng.time_linearize(20)
        local_obs = len(sh["obs_cam"])
        jac_bytes = BYTES_PER_OBS_JAC * local_obs
        jac_gbs = jac_bytes / (ms_jac * 1e-3) / 1e9
        nred = 6 * n_cams
        ms_factor, ms_bwd = st.cholesky_time_split(nred, reps=5)
        chol_flops = nred ** 3 / 3.0 + nred ** 2 / 2.0            # algorithmic flops of one n x n Cholesky
        chol_tflops = chol_flops / (ms_factor * 1e-3) / 1e12
        prof = st.cholesky_profile(nred)                          # stage-per-kernel schedule (diagnostic)
        syrk_tflops = prof["syrk_flops"] / (prof["ms_syrk"] * 1e-3) / 1e12 if prof["ms_syrk"] > 0 else 0.0
        # HBM bytes per launch from the PMC passes (tools/pmc_jacobian.sh -> profiles/pmc_jacobian.json);
        # only valid for the C5 shape it was collected on
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_jacobian.json")) as f:
                pj = json.load(f)
            if local_obs == 1000000:
                traffic = pj["hbm_bytes_per_launch"]
        except Exception:
            pass
        roof_jac = {"kernel": "ba_linearize_kernel<cams-in-LDS, with-Jacobian>", "bound": "hbm", "achieved": jac_gbs,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": jac_gbs / HBM_PEAK_GBS, "traffic": traffic,
                    "ms_per_launch": ms_jac, "algorithmic_bytes_per_launch": jac_bytes}
        chol_traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_chol.json")) as f:
                pc = json.load(f)
            if nred == 6000:
                chol_traffic = pc["hbm_bytes_per_launch"]      # PMC passes of tools/pmc_chol.sh, C5 shape only
        except Exception:
            pass
        roof_chol = {"kernel": "chol_mega_kernel (persistent dataflow Cholesky, v_mfma_f64_16x16x4_f64)",
                     "bound": "mfma", "achieved": chol_tflops, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": chol_tflops / FP64_MFMA_PEAK_TFLOPS, "traffic": chol_traffic, "ms_per_launch": ms_factor,
                     "algorithmic_flops_per_launch": chol_flops, "launches_per_lm_iteration": 1,
                     "microbench_ceiling": FP64_MFMA_MEASURED_CEILING_TFLOPS,
                     "frac_of_microbench_ceiling": chol_tflops / FP64_MFMA_MEASURED_CEILING_TFLOPS,
                     "note": "latency-bound: 47 dependent diagonal-block / panel hand-offs (DESIGN.md)"}
        # the dominant kernel by device time carries the headline roofline object
        out["roofline"] = roof_chol if ms_factor > ms_jac else roof_jac
        out["roofline_jacobian"] = roof_jac
        out["roofline_mfma"] = roof_chol
        out["cholesky_ms"] = {"factor_persistent_kernel": ms_factor, "backward": ms_bwd,
                              "stage_kernels_serial": {"diag": prof["ms_diag"], "trsm": prof["ms_trsm"],
                                                       "syrk": prof["ms_syrk"], "backward": prof["ms_bwd"],
                                                       "syrk_tflops": syrk_tflops}}

        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle_py as O       # cpu_baseline leg only: the oracle is the thing timed here
            ncpu = os.cpu_count() or 1

            def time_oracle(threads, budget_s):
                """fixed-work LM iterations of the SAME C5 problem on the host: 1 to calibrate, then as
                many more as fit in the budget (bounded sample)"""
                ba = O.BA(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
                tc = time.perf_counter()
                ba.solve(fixed_iterations=1, num_threads=threads)
                t1 = time.perf_counter() - tc
                extra = int(min(20, max(0, round(budget_s / max(t1, 1e-3)) - 1)))
                if extra > 0:
                    ba.solve(fixed_iterations=extra, num_threads=threads)
                dtc = time.perf_counter() - tc
                return (1 + extra) / dtc, 1 + extra, dtc
            # the reference pins num_threads = 1 (test_ceres.h:143); also report the best OpenMP setting
            # (measured on the GPU box's 2 x EPYC 9575F: 16 threads is the optimum of the oracle)
            it1, n1, d1 = time_oracle(1, 8.0)
            cands = sorted({min(ncpu, 16), min(ncpu, 32)})
            best = None
            for th in cands:
                r = time_oracle(th, 4.0)
                if best is None or r[0] > best[1][0]:
                    best = (th, r)
            thb, (itb, nb_, db) = best
            out["cpu_baseline"] = {"value": itb, "unit": "LM iterations/s", "cores": thb, "kind": "port",
                                   "residuals_per_sec": itb * 2.0 * n_obs,
                                   "sample": f"{nb_} fixed-work LM iterations of the same C5 problem, oracle/liboracle.so "
                                             f"(C port: OpenMP x{thb} of {ncpu} logical cores, dense Cholesky), {db:.1f} s wall",
                                   "seconds": db}
            out["cpu_baseline_single_thread"] = {"value": it1, "unit": "LM iterations/s", "cores": 1, "kind": "port",
                                                 "sample": f"{n1} iterations, num_threads = 1 as the reference pins "
                                                           f"(test_ceres.h:143), {d1:.1f} s wall", "seconds": d1}
            cpu_it = itb
            out["speedup_vs_cpu_port_single_thread"] = it_per_s / it1
            out["speedup_vs_cpu_port"] = it_per_s / cpu_it
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
