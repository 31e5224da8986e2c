"""Landmark sharding for the multi-GPU path.

One process per GPU.  Landmarks -- and with them their observations, which the reference's data
model already stores landmark-major (st20-g2o/src/include/sim_data.h:38-47) -- are split into
contiguous ranges balanced by observation count; the camera blocks are replicated.  Every rank
builds its partial reduced camera system locally; ONE cross-rank sum per build carries
[S | diag(Hcc) | gc | rhs | scalars]; every rank then factors the identical system redundantly
and back-substitutes its own landmarks.  A second, 4-double sum carries the trial-point cost
and step statistics.
"""
import numpy as np


def shard_cuts(obs_pt, n_pts, world):
    """contiguous landmark ranges [cuts[r], cuts[r+1]) with ~equal observation counts"""
    cnt = np.bincount(np.asarray(obs_pt), minlength=n_pts)
    csum = np.concatenate([[0], np.cumsum(cnt)])
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(csum, csum[-1] * r / world)))
    cuts.append(int(n_pts))
    return cuts


def make_shard(scene, rank, world):
    """this rank's sub-problem: all cameras, its landmarks (re-indexed from 0) and their observations"""
    n_pts = len(scene["pts0"])
    cuts = shard_cuts(scene["obs_pt"], n_pts, world)
    lo, hi = cuts[rank], cuts[rank + 1]
    m = (scene["obs_pt"] >= lo) & (scene["obs_pt"] < hi)
    return dict(cams0=scene["cams0"], pts0=scene["pts0"][lo:hi], obs_cam=scene["obs_cam"][m],
                obs_pt=(scene["obs_pt"][m] - lo).astype(np.int32), obs_feat=scene["obs_feat"][m],
                cam_fixed=scene["cam_fixed"], lo=lo, hi=hi)


class DeviceVector:
    """__cuda_array_interface__ view (FP64 vector) of a raw device pointer owned by the engine"""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def make_pg_shard(scene, rank, world):
    """pose graph (BASELINE C4): shard the EDGES into contiguous ranges, replicate the nodes"""
    m = len(scene["edge_i"])
    lo, hi = (m * rank) // world, (m * (rank + 1)) // world
    out = dict(scene)
    for k in ("edge_i", "edge_j", "meas"):
        out[k] = scene[k][lo:hi]
    out["lo"], out["hi"] = lo, hi
    return out


def torch_allreduce_hook(dist, torch):
    """all-reduce hook for BAEngine.set_allreduce: RCCL sum (torch.distributed, backend nccl) of the
    engine-owned device buffer, enqueued on the ENGINE's stream (the hook's `stream` argument, wrapped as
    a torch ExternalStream), so the collective is ordered with the kernels around it without events.
    Returns nonzero on any failure: a Python exception inside a ctypes callback would otherwise read as
    success and an unreduced system would be solved silently.  (bench.py uses the native communicator,
    slam-tricks_amd.Comm; this hook is the fallback and what the gloo CPU tests exercise.)"""
    def hook(_user, buf, count, stream):
        try:
            t = torch.as_tensor(DeviceVector(buf, count), device="cuda")
            if stream:
                with torch.cuda.stream(torch.cuda.ExternalStream(int(stream))):
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return 0
        except Exception as e:      # noqa: BLE001
            import sys
            print(f"stba all-reduce hook failed: {e!r}", file=sys.stderr, flush=True)
            return 1
    return hook
