"""host-only: how fast is LAPACK dpotrf (the OpenBLAS scipy ships) on this box at the reduced system's size?
usage: python tools/cpu_blas_check.py [n]"""
import sys, time
import numpy as np
from scipy.linalg import lapack
import threadpoolctl
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
rng = np.random.default_rng(0)
B = rng.standard_normal((n, n // 4))
A = B @ B.T + n * np.eye(n)
flops = n ** 3 / 3.0
for th in (8, 16, 32, 64, 128):
    with threadpoolctl.threadpool_limits(limits=th, user_api="blas"):
        best = 1e9
        for _ in range(3):
            M = np.asfortranarray(A.copy())
            t = time.perf_counter()
            c, info = lapack.dpotrf(M, lower=0, clean=0, overwrite_a=1)
            best = min(best, time.perf_counter() - t)
        print(f"dpotrf n={n} threads={th}: {best*1e3:.1f} ms, {flops/best/1e9:.0f} GFLOP/s", flush=True)
