import ctypes as C, sys, numpy as np
import torch
L = C.CDLL(sys.argv[1])
ms = np.zeros(4); fl = C.c_double(); flp = C.c_double(); nl = C.c_int()
for _ in range(2):
    rc = L.stba_cholesky_profile(6000, ms.ctypes.data_as(C.c_void_p), C.byref(fl), C.byref(flp), C.byref(nl), None)
print(sys.argv[1].split('/')[-1], "rc", rc, "diag %.3f trsm %.3f syrk %.3f bwd %.3f ms" % tuple(ms))
