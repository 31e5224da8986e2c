"""bench.py's launch path: `python bench.py --gpus N` must put N ranks on the job.  Here, without
GPUs: the self-launch under torch.distributed.run, the gloo rendezvous on 127.0.0.1, landmark sharding, the
cross-rank reduction and the single JSON line from rank 0.  The product has no CPU fallback, so the compute is
skipped (and says so) -- on the GPU box the same path times the sharded engines."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _env(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["TMPDIR"] = str(tmp_path)
    return env


def test_bench_gpus_2_launches_two_ranks(tmp_path):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--cams", "8",
                        "--pts", "200", "--obs-per-pt", "4", "--steps", "2", "--warmup", "0", "--no-cpu-baseline",
                        "--second-cams", "5", "--second-pts", "600"],
                       capture_output=True, text=True, timeout=600, env=_env(tmp_path), cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                      # ONE line, from rank 0
    d = json.loads(lines[0])
    st_has_gpu = "skipped" not in d
    assert d["n_gpus"] == 2 and d["ranks"] == 2
    if not st_has_gpu:
        assert d["value"] is None and "no CPU fallback" in d["skipped"]
        assert d["sharded_observations"] == d["n_obs"]    # the two landmark shards cover every observation once
    # what explains an N > 1 line: per-rank observation counts, the collective's time and bytes,
    # the phases as the maximum over ranks
    for key in ("observations_per_rank", "phase_ms_per_step", "allreduce_ms", "allreduce_bytes", "allreduce_calls_per_step"):
        assert key in d, key
    assert len(d["observations_per_rank"]) == 2 and sum(d["observations_per_rank"]) == d["n_obs"]
    assert min(d["observations_per_rank"]) > 0.4 * d["n_obs"]          # balanced by observation count
    # with N > 1 the LANDMARK-HEAVY scene is timed in the same call, under its own key: both shards cover it, and the
    # keys that make its line readable are there; every rank's camera blocks are compared bit for bit in-run
    assert "camera_blocks_identical_on_all_ranks" in d
    lh = d["landmark_heavy"]
    assert lh is not None and lh["n_cams"] == 5 and sum(lh["observations_per_rank"]) == lh["n_obs"] and len(lh["observations_per_rank"]) == 2
    for key in ("value", "predicted_scaling", "camera_blocks_identical_on_all_ranks"):
        assert key in lh, key
    if st_has_gpu:
        assert d["camera_blocks_identical_on_all_ranks"] is True and lh["camera_blocks_identical_on_all_ranks"] is True


def test_bench_refuses_a_rank_count_that_is_not_gpus(tmp_path):
    env = _env(tmp_path)
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--cams", "8", "--pts", "200",
                        "--obs-per-pt", "4"], capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)
