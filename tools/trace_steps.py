"""per-panel anatomy of a persistent-Cholesky trace: when D(b) became ready / started / ended, what made it ready last, and how busy
the workgroups were in between.  usage: python tools/trace_steps.py /tmp/mega_<tag>.bin"""
import sys
import numpy as np
raw = open(sys.argv[1], "rb").read()
nt = int(np.frombuffer(raw[:4], np.int32)[0])
tk = np.frombuffer(raw[4:4 + 16 * nt], np.int32).reshape(nt, 4).copy()
ty = tk[:, 0] & 0xff
tr = np.frombuffer(raw[4 + 16 * nt:4 + 16 * nt + 64 * nt], np.int64).reshape(nt, 8)
t0 = tr[:, 1].min()
us = lambda x: (x - t0) / 100.0
nwg = len(np.unique(tr[:, 0]))
D = {int(tk[k, 1]): k for k in range(nt) if ty[k] == 0}
TU = {}
for k in range(nt):
    if ty[k] == 5: TU.setdefault(int(tk[k, 1]), []).append(k)
print("  b | step  | D ticket->ready (parked) | D run | TU: first ticket, last ticket, last ready, last done (after D done) | busy fraction of the step")
prev_done = None
for b in sorted(D):
    k = D[b]
    done = us(tr[k, 3])
    s = f"{b:3d} | {done - prev_done:5.1f} |" if prev_done is not None else f"{b:3d} |   -   |"
    s += f" {us(tr[k, 2]) - us(tr[k, 1]):6.1f} | {done - us(tr[k, 2]):5.1f} |"
    if b in TU:
        ks = TU[b]
        s += f" {min(us(tr[q, 1]) for q in ks) - done:6.1f} {max(us(tr[q, 1]) for q in ks) - done:6.1f} {max(us(tr[q, 2]) for q in ks) - done:6.1f} {max(us(tr[q, 3]) for q in ks) - done:6.1f} |"
    if prev_done is not None:
        lo, hi = prev_done * 100 + t0, done * 100 + t0
        busy = (np.minimum(tr[:, 3], hi) - np.maximum(tr[:, 2], lo)).clip(0).sum() / ((hi - lo) * nwg)
        s += f" {busy:.2f}"
    prev_done = done
    print(s)
