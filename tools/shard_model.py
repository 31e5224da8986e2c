"""Design table for sharding the reduced camera system over the GPUs of one node (host only, no GPU):
the scheduling model of the persistent factorisation kernel with tile rows dealt block-cyclically to the GPUs
(include/stba.h, stba_cholesky_shard_model).   usage: python tools/shard_model.py > profiles/r2_shard_model.txt"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
st = importlib.import_module("slam-tricks_amd")
HOP_US, LINK_GBS, ROWS = 3.0, 48.0, 8
print(f"# model: 8 XCDs x 32 workgroups per GPU, task durations as measured on MI355X (dense_chol.hip, DUR), block-cyclic tile rows,")
print(f"# {ROWS} consecutive tile rows per GPU and round; cross-GPU dependency = {HOP_US} us flag hop + 128 KiB tile at {LINK_GBS} GB/s (one xGMI link,")
print(f"# one direction); link contention not modelled -- 'ingress' is the volume to hold against 7 links x ~48 GB/s")
print("# cameras unknowns | GPUs | makespan ms | speed-up | cross-GPU deps | ingress MiB (busiest GPU) | ingress ms at 300 GB/s")
for cams in (1000, 2000, 4000):
    n = 6 * cams
    base = st.cholesky_schedule_model(n)
    for g in (1, 2, 4, 8):
        ms, ce, ti = st.cholesky_shard_model(n, g, rows_per_group=ROWS, hop_us=HOP_US, link_gb_per_s=LINK_GBS)
        mib = ti * 128.0 / 1024.0
        print(f"{cams:6d} {n:8d} | {g:4d} | {ms/1e3:10.3f} | {base/ms:7.2f} | {int(ce):12d} | {mib:10.1f} | {mib*1.048576/300.0:8.3f}")
