"""N ranks on ONE device (rehearsal): do all ranks hold the same camera bits after k LM iterations?  Hook A: torch.distributed all_reduce
(gloo, CUDA tensor); hook B: all_gather + sum in rank order (the same bits on every rank by construction).
usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 tools/dbg/rank_bits.py [cams pts iters]"""
import hashlib, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.distributed as dist
st = importlib.import_module("slam-tricks_amd"); scenes = importlib.import_module("slam-tricks_amd.scenes"); sharding = importlib.import_module("slam-tricks_amd.sharding")
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
nc, npt, iters = (int(a) for a in (sys.argv[1:4] + ["60", "4000", "3"])[:3])
s = scenes.st20_scene(n_cams=nc, n_pts=npt, max_obs_per_pt=8, seed=5, pix_noise=1e-3)
sh = sharding.make_shard(s, rank, world)


def hook_gather(_u, buf, count, stream):
    t = torch.as_tensor(sharding.DeviceVector(buf, count), device="cuda")
    torch.cuda.synchronize()
    parts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    tot = parts[0].clone()
    for k in range(1, world): tot += parts[k]
    t.copy_(tot); torch.cuda.synchronize()
    return 0


for name, hook in (("all_reduce", sharding.torch_allreduce_hook(dist, torch)), ("gather+ordered sum", hook_gather)):
    e = st.BAEngine(sh["cams0"], sh["pts0"], sh["obs_cam"], sh["obs_pt"], sh["obs_feat"], sh["cam_fixed"])
    e.set_allreduce(hook, rank, world)
    e.lm_iterations(iters)
    cams, _ = e.get_params()
    dig = int.from_bytes(hashlib.sha1(np.ascontiguousarray(cams).tobytes()).digest()[:7], "little")
    t = torch.tensor([dig], dtype=torch.int64); out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    allc = [torch.zeros(cams.shape, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(allc, torch.from_numpy(cams))
    if rank == 0:
        md = max(float((allc[k] - allc[0]).abs().max()) for k in range(world))
        print(f"{name}: ranks identical {all(int(o) == int(out[0]) for o in out)}  max |cams_k - cams_0| {md:.3e}  timeouts {st.cholesky_timeout_count()}", flush=True)
dist.destroy_process_group()
