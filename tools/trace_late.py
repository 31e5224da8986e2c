"""late-phase anatomy of a persistent-Cholesky trace (tools/run_trace.sh): the short tasks' phases from panel B0 on, and the three
chains around one row.  usage: python tools/trace_late.py /tmp/mega_<tag>.bin [B0] [row]"""
import sys
import numpy as np
raw = open(sys.argv[1], "rb").read()
B0 = int(sys.argv[2]) if len(sys.argv) > 2 else 28
ROW = int(sys.argv[3]) if len(sys.argv) > 3 else 46
nt = int(np.frombuffer(raw[:4], np.int32)[0])
tk = np.frombuffer(raw[4:4 + 16 * nt], np.int32).reshape(nt, 4).copy()
ty = tk[:, 0] & 0xff
tr = np.frombuffer(raw[4 + 16 * nt:4 + 16 * nt + 64 * nt], np.int64).reshape(nt, 8)
t0 = tr[:, 1].min()
us = lambda x: (x - t0) / 100.0
names = ["D", "T", "TI", "U", "Uq", "TU"]
late = tk[:, 1] >= B0
for t, nph in ((0, 4), (1, 3), (5, 3), (4, 2)):
    m = (ty == t) & late & (tk[:, 2] < 47 if t == 1 else True)
    if not m.any(): continue
    wait = (tr[m, 2] - tr[m, 1]) / 100.0
    prev = tr[m, 2]; out = []
    for k in range(nph):
        out.append(np.median((tr[m, 4 + k] - prev) / 100.0)); prev = tr[m, 4 + k]
    out.append(np.median((tr[m, 3] - prev) / 100.0))
    print(f"{names[t]:3s} panels >= {B0}: n {m.sum():4d}  parked median {np.median(wait):6.2f}  phases (median us) {np.round(out, 2)}  run median {np.median((tr[m,3]-tr[m,2])/100.0):.2f}")
idx = {}
for k in range(nt): idx[(int(ty[k]), int(tk[k, 1]), int(tk[k, 2]), int(tk[k, 3]))] = k
print(f"row {ROW}:  b | D done | T(b;row) ready done | U/Uq(b;row,b+1) ready done | TU(b) done (max of 4)")
for b in range(B0, 45):
    d = idx.get((0, b, 0, 0)); t = idx.get((1, b, ROW, 0))
    u = idx.get((3, b, ROW, b + 1))
    uq = [idx.get((4, b, ROW * 4 + q, b + 1)) for q in range(4)]
    tu = [idx.get((5, b, q, 0)) for q in range(4)]
    s = f"{b:3d} | {us(tr[d,3]):8.1f} | "
    s += f"{us(tr[t,2]):8.1f} {us(tr[t,3]):8.1f} | " if t is not None else "   -      -    | "
    if u is not None: s += f"{us(tr[u,2]):8.1f} {us(tr[u,3]):8.1f} | "
    elif uq[0] is not None: s += f"{min(us(tr[q,2]) for q in uq):8.1f} {max(us(tr[q,3]) for q in uq):8.1f}q| "
    else: s += "   -      -    | "
    if tu[0] is not None: s += f"{max(us(tr[q,3]) for q in tu):8.1f}"
    print(s)
