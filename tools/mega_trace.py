"""task timeline of the persistent Cholesky kernel.
usage: STBA_MEGA_TRACE=/tmp/mega.bin python tools/mega_trace.py run [n];  python tools/mega_trace.py /tmp/mega.bin"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
    st = importlib.import_module("slam-tricks_amd")
    print("chol ms", st.cholesky_time(n, reps=2))
    sys.exit(0)
raw = open(sys.argv[1], "rb").read()
nt = int(np.frombuffer(raw[:4], np.int32)[0])
tasks = np.frombuffer(raw[4:4 + 16 * nt], np.int32).reshape(nt, 4).copy()
tasks[:, 0] &= 0xff
tr = np.frombuffer(raw[4 + 16 * nt:4 + 16 * nt + 64 * nt], np.int64).reshape(nt, 8)
sim = np.frombuffer(raw[4 + 80 * nt:4 + 84 * nt], np.float32) if len(raw) >= 4 + 84 * nt else None
t0 = tr[:, 1].min()
us = lambda x: (x - t0) / 100.0
names = ["D", "T", "TI", "U", "Uq", "TU"]
print("tasks", nt, "span us", us(tr[:, 3].max()))
for ty in range(6):
    m = tasks[:, 0] == ty
    if m.any():
        wait = (tr[m, 2] - tr[m, 1]) / 100.0
        run = (tr[m, 3] - tr[m, 2]) / 100.0
        print(f"{names[ty]:3s} n={m.sum():6d} wait mean {wait.mean():7.2f} us (sum {wait.sum()/1e3:8.2f} ms)  run mean {run.mean():7.2f} med {np.median(run):7.2f} max {run.max():7.2f} (sum {run.sum()/1e3:8.2f} ms)")
for ty, nph in ((0, 4), (1, 3), (2, 3), (4, 2), (5, 3)):
    m = tasks[:, 0] == ty
    if m.any():
        prev = tr[m, 2]
        out = []
        for k in range(nph):
            out.append(((tr[m, 4 + k] - prev) / 100.0).mean()); prev = tr[m, 4 + k]
        out.append(((tr[m, 3] - prev) / 100.0).mean())
        print(f"{names[ty]:3s} phases (us):", np.round(out, 2))
bulk = tr[:, 0] >= 100000          # tasks run by the 256-thread bulk class (two workgroups per CU)
for nm, m in (("chain / only class", ~bulk), ("bulk class", bulk)):
    if not m.any(): continue
    nwg = len(np.unique(tr[m, 0]))
    busy = (tr[m, 3] - tr[m, 2]).sum() / 100.0
    print(f"{nm}: workgroups {nwg} busy fraction {busy / (nwg * us(tr[:, 3].max())):.3f}")
    mu = m & (tasks[:, 0] == 3)
    if mu.any():
        nbp = np.maximum(1, (np.frombuffer(raw[4:4 + 16 * nt], np.int32).reshape(nt, 4)[:, 0] >> 16) & 0xff)
        for k in (1, 2):
            mk = mu & (nbp == k)
            if mk.any():
                run = (tr[mk, 3] - tr[mk, 2]) / 100.0
                print(f"   U tasks of {k} panel(s): n {mk.sum()} run mean {run.mean():.2f} med {np.median(run):.2f} p10 {np.percentile(run, 10):.2f} p90 {np.percentile(run, 90):.2f}")
# critical chain: D(b) start/end
dm = np.where(tasks[:, 0] == 0)[0]
print("  b   D.ready   D.done   (ready - prev done)")
prev = None
for k in dm[:: max(1, len(dm) // 24)]:
    b = tasks[k, 1]
    print(f"{b:3d} {us(tr[k,2]):9.1f} {us(tr[k,3]):9.1f}")
# per step: time between D(b) done and D(b+1) ready
d_done = {int(tasks[k, 1]): us(tr[k, 3]) for k in dm}
d_ready = {int(tasks[k, 1]): us(tr[k, 2]) for k in dm}
gaps = [d_ready[b + 1] - d_done[b] for b in range(len(dm) - 1)]
runs = [d_done[b] - d_ready[b] for b in range(len(dm))]
print("D run mean", np.mean(runs), " gap D(b).done -> D(b+1).ready: mean", np.mean(gaps), "first10", np.round(gaps[:10], 1), "last10", np.round(gaps[-10:], 1))
# hand-off detail for a few steps: times relative to D(b).done
for b in (2, 5, 10, 20, 30, 40):
    if b + 1 >= len(dm):
        continue
    kd = [k for k in dm if tasks[k, 1] == b][0]
    kd1 = [k for k in dm if tasks[k, 1] == b + 1][0]
    t0b = tr[kd, 3]
    rel = lambda x: (x - t0b) / 100.0
    tus = [k for k in np.where(tasks[:, 0] == 5)[0] if tasks[k, 1] == b]
    line = f"b={b:2d} D.run {(tr[kd,3]-tr[kd,2])/100.0:5.1f} | "
    for k in tus:
        line += f"TU{tasks[k,2]} tick {rel(tr[k,1]):7.1f} rdy {rel(tr[k,2]):6.1f} ph {rel(tr[k,4]):5.1f} {rel(tr[k,5]):5.1f} {rel(tr[k,6]):5.1f} done {rel(tr[k,3]):5.1f} | "
    line += f"D+1 tick {rel(tr[kd1,1]):7.1f} rdy {rel(tr[kd1,2]):6.1f}"
    print(line)
# full chain table: every time relative to D(b).done
print("\nchain table (us relative to D(b).done):  step = D(b+1).done - D(b).done")
print("  b  step | TU tick  rdy  done | Ubb tick  rdy  done | Uq/U(b-1;b+1) tick rdy done | T(b-1,b+1) tick rdy done | D+1 tick rdy")
key = {}
for k in range(nt):
    key[(int(tasks[k, 0]), int(tasks[k, 1]), int(tasks[k, 2]), int(tasks[k, 3]))] = k
for b in range(1, len(dm) - 1):
    kd = key[(0, b, 0, 0)]; kd1 = key[(0, b + 1, 0, 0)]
    t0b = tr[kd, 3]
    rel = lambda x: (x - t0b) / 100.0
    tus = [key[(5, b, q, 0)] for q in range(4)]
    ubb = key.get((3, b - 1, b + 1, b + 1))
    uqs = [key[(4, b - 1, (b + 1) * 4 + q, b)] for q in range(4) if (4, b - 1, (b + 1) * 4 + q, b) in key]
    tk = key.get((1, b - 1, b + 1, 0))
    f = lambda ks, c, fn: fn(rel(tr[k, c]) for k in ks)
    line = f"{b:3d} {(tr[kd1,3]-tr[kd,3])/100.0:5.1f} | {f(tus,1,max):6.1f} {f(tus,2,max):6.1f} {f(tus,3,max):6.1f} | "
    if ubb is not None:
        line += f"{rel(tr[ubb,1]):6.1f} {rel(tr[ubb,2]):6.1f} {rel(tr[ubb,3]):6.1f} | "
    if uqs:
        line += f"{f(uqs,1,max):6.1f} {f(uqs,2,max):6.1f} {f(uqs,3,max):6.1f} | "
    if tk is not None:
        line += f"{rel(tr[tk,1]):6.1f} {rel(tr[tk,2]):6.1f} {rel(tr[tk,3]):6.1f} | "
    line += f"{rel(tr[kd1,1]):6.1f} {rel(tr[kd1,2]):6.1f}"
    print(line)
# queue neighbourhood of the late diagonal-tile updates: who sat in front of them?
if len(sys.argv) > 2:
    for b in [int(x) for x in sys.argv[2].split(",")]:
        k = key.get((1, b - 1, b + 1, 0)) if os.environ.get("TRACE_T") else key.get((3, b - 1, b + 1, b + 1))
        if k is None:
            continue
        t0b = tr[key[(0, b, 0, 0)], 3]
        rel = lambda x: (x - t0b) / 100.0
        print(f"\nqueue in front of U({b-1};{b+1},{b+1}) (times rel. D({b}).done): idx type b i j | wg tick rdy done | wait")
        for kk in range(max(0, k - 70), k + 3):
            ty, bb, ii, jj = tasks[kk]
            print(f"{kk:6d} {names[ty]:3s} {bb:3d} {ii:4d} {jj:3d} | {tr[kk,0]:4d} {rel(tr[kk,1]):7.1f} {rel(tr[kk,2]):7.1f} {rel(tr[kk,3]):7.1f} | {(tr[kk,2]-tr[kk,1])/100.0:6.1f}")

# drift of the machine against the host's simulation: (real start - simulated start) per 200 us of simulated time
if sim is not None:
    real = us(tr[:, 2])
    print("\nsim window(us)  mean real-sim (us) by task class:   D/TU      T     U(j<=b+2)   U(far)")
    cls = np.where(np.isin(tasks[:, 0], (0, 5)), 0, np.where(tasks[:, 0] == 1, 1, np.where((tasks[:, 0] == 4) | (tasks[:, 3] <= tasks[:, 1] + 2), 2, 3)))
    for a0 in np.arange(0, sim.max() + 200, 200.0):
        m = (sim >= a0) & (sim < a0 + 200)
        if not m.any(): continue
        out = []
        for c in range(4):
            mm = m & (cls == c)
            out.append(f"{(real[mm] - sim[mm]).mean():8.1f}" if mm.any() else "       -")
        print(f"{a0:6.0f}-{a0+200:6.0f}   " + "  ".join(out))
    print("simulated makespan", sim.max(), "real", real.max())
if sim is not None:
    ds = {int(tasks[k, 1]): float(sim[k]) for k in dm}
    print("simulated step D(b+1).start - D(b).start:", np.round([ds[b + 1] - ds[b] for b in range(len(dm) - 1)], 1))
    # what delayed the simulated TU / D?  inputs' simulated end times relative to D(b) end
    DURS = {0: 23.0, 1: 23.0, 2: 19.0, 3: 25.0, 4: 16.5, 5: 20.0}
    for b in range(1, len(dm) - 1):
        if ds[b + 1] - ds[b] < 52: continue
        dend = ds[b] + 23.0
        tu = [key[(5, b, q, 0)] for q in range(4)]
        ubb = key.get((3, b - 1, b + 1, b + 1))
        uqs = [key[(4, b - 1, (b + 1) * 4 + q, b)] for q in range(4) if (4, b - 1, (b + 1) * 4 + q, b) in key]
        tk = key.get((1, b - 1, b + 1, 0))
        print(f"  sim b={b}: step {ds[b+1]-ds[b]:.1f}  TU start {max(sim[k] for k in tu)-dend:6.1f}  Ubb start {sim[ubb]-dend:6.1f}  Uq start {max(sim[k] for k in uqs)-dend:6.1f}  T(b-1,b+1) start {sim[tk]-dend:6.1f}  (D(b-1) end {ds[b-1]+23-dend:6.1f})")
