"""distribution of ticket -> ready times of the trailing-update tasks in the busy part of a persistent-Cholesky trace
usage: python tools/trace_waits.py /tmp/mega_<tag>.bin [b_lo b_hi]"""
import sys
import numpy as np
raw = open(sys.argv[1], "rb").read()
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3, 25)
nt = int(np.frombuffer(raw[:4], np.int32)[0])
tk = np.frombuffer(raw[4:4 + 16 * nt], np.int32).reshape(nt, 4).copy()
ty = tk[:, 0] & 0xff
tr = np.frombuffer(raw[4 + 16 * nt:4 + 16 * nt + 64 * nt], np.int64).reshape(nt, 8)
m = (ty == 3) & (tk[:, 1] >= lo) & (tk[:, 1] <= hi)
w = (tr[m, 2] - tr[m, 1]) / 100.0
r = (tr[m, 3] - tr[m, 2]) / 100.0
print(f"U tasks of panels {lo}..{hi}: n {m.sum()}  ticket->ready: p10 {np.percentile(w,10):.2f} p25 {np.percentile(w,25):.2f} median {np.median(w):.2f} p75 {np.percentile(w,75):.2f} p90 {np.percentile(w,90):.2f} mean {w.mean():.2f};  run median {np.median(r):.2f}")
print(f"   share of workgroup time: waiting {w.sum() / (w.sum() + r.sum()):.3f}; waits under 4 us: {np.mean(w < 4):.2f} of the tasks, {w[w < 4].sum() / w.sum():.2f} of the waiting time")
# gap between the end of a workgroup's task and its next ticket poll
order = np.lexsort((tr[:, 2], tr[:, 0]))
wg = tr[order, 0]; t_poll = tr[order, 1]; t_done = tr[order, 3]; tyo = ty[order]; bo = tk[order, 1]
same = wg[1:] == wg[:-1]
gap = (t_poll[1:] - t_done[:-1])[same & (tyo[1:] == 3) & (bo[1:] >= lo) & (bo[1:] <= hi)] / 100.0
print(f"   previous task's end -> this ticket's poll start: median {np.median(gap):.2f} p90 {np.percentile(gap, 90):.2f} us")
