"""Design study for sharding the reduced camera system ("what comes next"; DESIGN.md "sharded reduced
solve").  Nothing here runs on a GPU:

* the host-side scheduling model of the persistent factorisation kernel (the code that orders the kernel's tickets)
  with the tile rows dealt block-cyclically to several GPUs and a price on every dependency that crosses GPUs;
* the message protocol of that distribution -- the owner of tile row b factors the diagonal block and broadcasts it,
  every owner solves its panel tiles and broadcasts them, every owner updates its own rows -- executed by two gloo
  ranks with numpy tiles and the library's ownership map, against numpy's Cholesky.
"""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

st = importlib.import_module("slam-tricks_amd")


def test_one_gpu_shard_model_is_the_production_model():
    for n in (1500, 6000):
        ms, cross, tiles = st.cholesky_shard_model(n, 1)
        assert ms == st.cholesky_schedule_model(n)
        assert cross == 0 and tiles == 0


def test_shard_model_says_when_sharding_pays():
    # C5's reduced system (6000 unknowns) is bound by the dependent chain of diagonal blocks: more GPUs buy ~10 %
    base6 = st.cholesky_schedule_model(6000)
    ms8, cross8, tiles8 = st.cholesky_shard_model(6000, 8, rows_per_group=8)
    assert 1.0 < base6 / ms8 < 1.3
    # four times the cameras: the trailing updates dominate and eight GPUs are worth 4x or more
    base24 = st.cholesky_schedule_model(24000)
    ms24, cross24, tiles24 = st.cholesky_shard_model(24000, 8, rows_per_group=8)
    assert base24 / ms24 > 4.0
    ms24_2, _, _ = st.cholesky_shard_model(24000, 2, rows_per_group=8)
    assert ms24 < ms24_2 < base24
    # every panel tile is fetched once by each GPU that does not own it: at most (G-1)/G of the nblk^2/2 tiles
    nblk = (24000 + 1 + 127) // 128
    assert 0 < tiles24 <= nblk * (nblk + 1) / 2
    assert cross24 > tiles24          # (a tile feeds many remote updates but crosses once per GPU)


def test_shard_model_prices_the_hops():
    a, _, _ = st.cholesky_shard_model(6000, 4, rows_per_group=8, hop_us=0.0, link_gb_per_s=1e6)
    b, _, _ = st.cholesky_shard_model(6000, 4, rows_per_group=8, hop_us=3.0, link_gb_per_s=48.0)
    c, _, _ = st.cholesky_shard_model(6000, 4, rows_per_group=8, hop_us=30.0, link_gb_per_s=48.0)
    assert a < b < c
    with pytest.raises(st.StbaError):
        st.cholesky_shard_model(6000, 0)
    with pytest.raises(st.StbaError):
        st.cholesky_shard_model(6000, 2, link_gb_per_s=0.0)


def test_block_cyclic_owner_map():
    own = st.cholesky_shard_owner(20, 2, 4)
    assert own.tolist() == [0] * 4 + [1] * 4 + [0] * 4 + [1] * 4 + [0] * 4
    own = st.cholesky_shard_owner(7, 3, 1)
    assert own.tolist() == [0, 1, 2, 0, 1, 2, 0]
    with pytest.raises(st.StbaError):
        st.cholesky_shard_owner(4, 2, 0)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _protocol_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    lib = importlib.import_module("slam-tricks_amd")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    T, nblk, R = 8, 11, 2                  # tile size, tile rows, rows per group
    n = T * nblk
    rng = np.random.default_rng(21)        # (the same matrix on every rank; each rank only TOUCHES the rows it owns)
    B = rng.standard_normal((n, n))
    A = B @ B.T + n * np.eye(n)
    own = lib.cholesky_shard_owner(nblk, world, R)
    tile = lambda M, i, j: M[i * T:(i + 1) * T, j * T:(j + 1) * T]
    mine = [i for i in range(nblk) if own[i] == rank]
    W = np.full_like(A, np.nan)            # this rank's rows of the trailing matrix / of L
    for i in mine:
        W[i * T:(i + 1) * T, :(i + 1) * T] = A[i * T:(i + 1) * T, :(i + 1) * T]
    sent = 0
    for b in range(nblk):
        # D(b): the owner of row b factors the diagonal tile and broadcasts it
        Lbb = torch.zeros(T, T, dtype=torch.float64)
        if own[b] == rank:
            Lbb = torch.from_numpy(np.linalg.cholesky(tile(W, b, b)))
            tile(W, b, b)[:] = Lbb.numpy()
        dist.broadcast(Lbb, src=int(own[b]))
        # T(b; i): every owner solves its panel tiles ...
        for i in mine:
            if i > b:
                tile(W, i, b)[:] = np.linalg.solve(Lbb.numpy(), tile(W, i, b).T).T
        # ... and broadcasts them (the model's "remote tiles": one transfer per tile and remote GPU)
        panel = {}
        for j in range(b + 1, nblk):
            t = torch.from_numpy(tile(W, j, b).copy()) if own[j] == rank else torch.zeros(T, T, dtype=torch.float64)
            dist.broadcast(t, src=int(own[j]))
            panel[j] = t.numpy()
            sent += int(own[j] == rank)
        # U(b; i, j): every owner updates its own rows with its own tile and the (mostly remote) tile of row j
        for i in mine:
            for j in range(b + 1, i + 1):
                tile(W, i, j)[:] -= panel[i] @ panel[j].T
    # collect L on rank 0
    out = torch.from_numpy(np.nan_to_num(np.tril(W), nan=0.0))
    dist.reduce(out, dst=0)
    if rank == 0:
        L = out.numpy()
        q.put((float(np.abs(L - np.linalg.cholesky(A)).max()), sent))
    else:
        q.put((0.0, sent))
    dist.destroy_process_group()


def test_two_rank_block_cyclic_factorisation_protocol():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_protocol_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert max(e for e, _ in res) < 1e-10
    # every panel tile left its owner exactly once: sum over panels of the tiles below the diagonal
    assert sum(s for _, s in res) == 11 * 10 // 2
