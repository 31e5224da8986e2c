"""chain B of a persistent-Cholesky trace, per panel b (late phase): what makes TU(b) ready.
usage: python tools/trace_chainb.py /tmp/mega_<tag>.bin [B0]"""
import sys
import numpy as np
raw = open(sys.argv[1], "rb").read()
B0 = int(sys.argv[2]) if len(sys.argv) > 2 else 30
nt = int(np.frombuffer(raw[:4], np.int32)[0])
tk = np.frombuffer(raw[4:4 + 16 * nt], np.int32).reshape(nt, 4).copy()
ty = tk[:, 0] & 0xff
tr = np.frombuffer(raw[4 + 16 * nt:4 + 16 * nt + 64 * nt], np.int64).reshape(nt, 8)
t0 = tr[:, 1].min()
us = lambda x: (x - t0) / 100.0
by = {}
for k in range(nt): by.setdefault((int(ty[k]), int(tk[k, 1])), []).append(k)
print("  b | D(b) done | relative to it:  TU(b-1) all done | T(b-1;b+1) parts done | Uq(b-1;b+1,.,b) ready .. done | U(b-2..;b+1,b) done | TU(b) last ready, last done | D(b+1) ready")
for b in range(B0, 46):
    d = by[(0, b)][0]; dd = us(tr[d, 3])
    tu_prev = [k for k in by.get((5, b - 1), [])]
    tparts = [k for k in by.get((1, b - 1), []) if tk[k, 2] == b + 1]
    uq = [k for k in by.get((4, b - 1), []) if (tk[k, 2] >> 2) == b + 1 and tk[k, 3] == b]
    uw = [k for k in by.get((3, b - 1), []) if tk[k, 2] == b + 1 and tk[k, 3] == b]
    tu = by.get((5, b), [])
    d1 = by.get((0, b + 1), [None])[0]
    f = lambda ks, col, fn: (f"{fn(us(tr[k, col]) for k in ks) - dd:7.1f}" if ks else "   -   ")
    print(f"{b:3d} | {dd:8.1f} | {f(tu_prev, 3, max)} | {f(tparts, 3, max)} | {f(uq or uw, 2, min)} .. {f(uq or uw, 3, max)} | {f(tu, 2, max)} {f(tu, 3, max)} | " + (f"{us(tr[d1, 2]) - dd:7.1f}" if d1 is not None else ""))
