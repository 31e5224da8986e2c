"""Library cross-check of the reduced solve ("rocSOLVER potrf is the cross-check, not the deliverable"):
times rocsolver_dpotrf + rocsolver_dpotrs (one right-hand side) on the same box next to stba's own factorisation.
librocsolver is dlopen'ed HERE, in a tool; libstba.so never links or calls it.  A stated baseline, never a fallback.
usage: python tools/rocsolver_potrf.py [n ...]   -> one JSON line per size on stdout"""
import ctypes as C, importlib, json, os, sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _load():
    for name in ("librocsolver.so.0", "librocsolver.so", "/opt/rocm/lib/librocsolver.so"):
        try:
            rs = C.CDLL(name, mode=C.RTLD_GLOBAL)
            break
        except OSError:
            rs = None
    for name in ("librocblas.so.5", "librocblas.so.4", "librocblas.so", "/opt/rocm/lib/librocblas.so"):
        try:
            rb = C.CDLL(name, mode=C.RTLD_GLOBAL)
            break
        except OSError:
            rb = None
    return rs, rb


def time_potrf(n, reps=5):
    rs, rb = _load()
    if rs is None or rb is None:
        return {"n": n, "found": False}
    dev = torch.device("cuda:0")
    h = C.c_void_p()
    assert rb.rocblas_create_handle(C.byref(h)) == 0
    st = torch.cuda.current_stream().cuda_stream
    assert rb.rocblas_set_stream(h, C.c_void_p(st)) == 0
    g = torch.Generator(device=dev); g.manual_seed(7)
    # SPD with a bounded condition number, built in slabs so that n = 24 000 fits comfortably
    M = torch.rand((n, 64), generator=g, device=dev, dtype=torch.float64)
    S = M @ M.T
    S += torch.eye(n, device=dev, dtype=torch.float64) * 64.0
    b = torch.rand((n,), generator=g, device=dev, dtype=torch.float64)
    A = torch.empty_like(S)
    x = torch.empty_like(b)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    LOWER = 122
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf, ts = [], []
    for k in range(reps + 2):
        A.copy_(S); x.copy_(b)
        e[0].record()
        rc = rs.rocsolver_dpotrf(h, LOWER, n, C.c_void_p(A.data_ptr()), n, C.c_void_p(info.data_ptr()))
        e[1].record()
        rc2 = rs.rocsolver_dpotrs(h, LOWER, n, 1, C.c_void_p(A.data_ptr()), n, C.c_void_p(x.data_ptr()), n)
        e[2].record()
        torch.cuda.synchronize()
        assert rc == 0 and rc2 == 0 and int(info.item()) == 0, (rc, rc2, int(info.item()))
        if k >= 2:
            tf.append(e[0].elapsed_time(e[1])); ts.append(e[1].elapsed_time(e[2]))
    resid = float(((S @ x - b).abs().max() / b.abs().max()).item())
    rb.rocblas_destroy_handle(h)
    tf.sort(); ts.sort()
    out = {"n": n, "found": True, "potrf_ms_median": tf[len(tf) // 2], "potrf_ms_min": tf[0], "potrs_ms_median": ts[len(ts) // 2],
           "potrf_tflops": (n ** 3 / 3.0) / tf[len(tf) // 2] / 1e9, "solve_residual_rel": resid}
    try:
        stm = importlib.import_module("slam-tricks_amd")
        f, bwd = stm.cholesky_time_split(n, reps=reps)
        out["stba_factor_ms"] = f; out["stba_backward_ms"] = bwd
        out["stba_tflops"] = (n ** 3 / 3.0) / f / 1e9
        out["rocsolver_over_stba"] = (out["potrf_ms_median"] + out["potrs_ms_median"]) / (f + bwd)
    except Exception as ex:      # the tool still reports the library's time
        out["stba_error"] = str(ex)
    del S, A, M
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    for n in [int(a) for a in sys.argv[1:]] or [6000, 12000, 24000]:
        print(json.dumps(time_potrf(n)), flush=True)
