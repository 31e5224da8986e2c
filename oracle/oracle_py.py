"""ctypes view of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/oracle.h).  The product never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

TRACE_COLS = 7
TERM_REASON = {0: "none", 1: "gradient", 2: "function", 3: "parameter", 4: "max_iter",
               5: "min_radius", 6: "solver_fail", 7: "fixed"}


class LMOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int),
                ("initial_trust_region_radius", C.c_double),
                ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double),
                ("min_relative_decrease", C.c_double),
                ("min_lm_diagonal", C.c_double),
                ("max_lm_diagonal", C.c_double),
                ("function_tolerance", C.c_double),
                ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double),
                ("jacobi_scaling", C.c_int),
                ("num_threads", C.c_int),
                ("fixed_iterations", C.c_int),
                ("function_tolerance_takes_step", C.c_int)]


class LMSummary(C.Structure):
    _fields_ = [("termination_type", C.c_int),
                ("termination_reason", C.c_int),
                ("num_iterations", C.c_int),
                ("num_successful_steps", C.c_int),
                ("num_unsuccessful_steps", C.c_int),
                ("initial_cost", C.c_double),
                ("final_cost", C.c_double),
                ("final_radius", C.c_double),
                ("final_gradient_max_norm", C.c_double),
                ("seconds_total", C.c_double),
                ("seconds_linearize", C.c_double),
                ("seconds_schur", C.c_double),
                ("seconds_solve", C.c_double),
                ("seconds_backsub", C.c_double),
                ("seconds_cost", C.c_double)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["termination_reason"] = TERM_REASON.get(d["termination_reason"], "?")
        return d


class BAProblem(C.Structure):
    _fields_ = [("n_cams", C.c_int), ("n_pts", C.c_int), ("n_obs", C.c_int),
                ("cams", C.c_void_p), ("pts", C.c_void_p),
                ("obs_cam", C.c_void_p), ("obs_pt", C.c_void_p), ("obs_feat", C.c_void_p),
                ("cam_fixed", C.c_void_p), ("pt_fixed", C.c_void_p)]


def build(force=False):
    """Compile liboracle.so (gcc).  Building the checker is not using it."""
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("oracle.c", "oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_ba_evaluate.restype = C.c_double
        _LIB.orc_calib_evaluate.restype = C.c_double
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


_DENSE_CB = None      # (keeps the ctypes callback alive)


def use_lapack(on=True, threads=None):
    """bench.py's cpu_baseline leg: let the LM solve of the oracle factor the reduced camera system with LAPACK
    (dpotrf / dpotrs of the OpenBLAS that scipy ships) instead of the blocked C loop of orc_cholesky_lower --
    the fair CPU opponent for a dense 6000 x 6000 FP64 Cholesky.  Returns a description of the BLAS in use, or None
    if scipy's LAPACK is not importable.  use_lapack(False) restores the C factorisation."""
    global _DENSE_CB
    L = lib()
    L.orc_set_dense_solver.argtypes = [C.c_void_p]
    if not on:
        L.orc_set_dense_solver(None)
        _DENSE_CB = None
        return None
    try:
        from scipy.linalg import lapack
    except Exception:
        return None
    info = "scipy LAPACK"
    try:
        import threadpoolctl
        if threads:
            threadpoolctl.threadpool_limits(limits=int(threads), user_api="blas")
        for d in threadpoolctl.threadpool_info():
            if d.get("user_api") == "blas" and "scipy" in os.path.basename(d.get("filepath", "")) and "scipy.libs" in d.get("filepath", ""):
                info = f"{d.get('internal_api')} {d.get('version')} ({d.get('architecture')}), {d.get('num_threads')} threads"
    except Exception:
        pass

    def solve(S_ptr, n, x_ptr):
        # the row-major lower triangle is the column-major UPPER triangle of the same memory: factor S^T = U^T U in place
        S = np.ctypeslib.as_array(C.cast(S_ptr, C.POINTER(C.c_double)), shape=(n, n)).T
        x = np.ctypeslib.as_array(C.cast(x_ptr, C.POINTER(C.c_double)), shape=(n,))
        c, bad = lapack.dpotrf(S, lower=0, clean=0, overwrite_a=1)
        if bad != 0:
            return int(bad) if bad > 0 else n
        sol, bad = lapack.dpotrs(c, x, lower=0)
        if bad != 0:
            return n
        x[:] = sol
        return 0

    _DENSE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p)(solve)
    L.orc_set_dense_solver(C.cast(_DENSE_CB, C.c_void_p))
    return info


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def default_options(**kw):
    o = LMOptions()
    lib().orc_lm_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


# ---------------------------------------------------------------- SO3 helpers
def so3_exp(w):
    q = np.zeros(4)
    lib().orc_so3_exp(_p(f64(w)), _p(q))
    return q


def so3_log(q):
    w = np.zeros(3)
    lib().orc_so3_log(_p(f64(q)), _p(w))
    return w


def so3_plus(q, d):
    o = np.zeros(4)
    lib().orc_so3_plus(_p(f64(q)), _p(f64(d)), _p(o))
    return o


def so3_plus_jacobian(q):
    J = np.zeros(12)
    lib().orc_so3_plus_jacobian(_p(f64(q)), _p(J))
    return J.reshape(4, 3)


def so3r3_plus(x, d):
    o = np.zeros(3)
    lib().orc_so3r3_plus(_p(f64(x)), _p(f64(d)), _p(o))
    return o


def quat_to_rot(q):
    R = np.zeros(9)
    lib().orc_quat_to_rot(_p(f64(q)), _p(R))
    return R.reshape(3, 3)


def rot_to_quat(R):
    q = np.zeros(4)
    lib().orc_rot_to_quat(_p(f64(R).reshape(-1)), _p(q))
    return q


def se3_exp(xi):
    q, t = np.zeros(4), np.zeros(3)
    lib().orc_se3_exp(_p(f64(xi)), _p(q), _p(t))
    return q, t


def se3_log(q, t):
    xi = np.zeros(6)
    lib().orc_se3_log(_p(f64(q)), _p(f64(t)), _p(xi))
    return xi


# ---------------------------------------------------------------- reprojection factor
def reproj_residual(q, t, L, f, ambient=False):
    r = np.zeros(2)
    fn = lib().orc_reproj_residual_ambient if ambient else lib().orc_reproj_residual
    fn(_p(f64(q)), _p(f64(t)), _p(f64(L)), _p(f64(f)), _p(r))
    return r


def reproj_jacobian(q, t, L, rot_mode=0):
    Jc, Jp = np.zeros(12), np.zeros(6)
    lib().orc_reproj_jacobian(_p(f64(q)), _p(f64(t)), _p(f64(L)), _p(Jc), _p(Jp), C.c_int(rot_mode))
    return Jc.reshape(2, 6), Jp.reshape(2, 3)


# ---------------------------------------------------------------- bundle adjustment
class BA:
    """Holds numpy arrays alive and exposes the oracle BA entry points."""

    def __init__(self, cams, pts, obs_cam, obs_pt, obs_feat, cam_fixed=None, pt_fixed=None):
        self.cams = f64(cams).copy().reshape(-1, 7)
        self.pts = f64(pts).copy().reshape(-1, 3)
        self.obs_cam = np.ascontiguousarray(obs_cam, dtype=np.int32)
        self.obs_pt = np.ascontiguousarray(obs_pt, dtype=np.int32)
        assert np.all(np.diff(self.obs_pt) >= 0), "oracle wants landmark-major observations"
        self.obs_feat = f64(obs_feat).reshape(-1, 2)
        self.cam_fixed = None if cam_fixed is None else np.ascontiguousarray(cam_fixed, dtype=np.uint8).reshape(-1, 6)
        self.pt_fixed = None if pt_fixed is None else np.ascontiguousarray(pt_fixed, dtype=np.uint8)
        self.nc, self.np_, self.no = len(self.cams), len(self.pts), len(self.obs_cam)

    def _struct(self):
        return BAProblem(self.nc, self.np_, self.no, _p(self.cams), _p(self.pts), _p(self.obs_cam),
                         _p(self.obs_pt), _p(self.obs_feat), _p(self.cam_fixed), _p(self.pt_fixed))

    def evaluate(self, jac=True):
        r = np.zeros((self.no, 2))
        Jc = np.zeros((self.no, 2, 6)) if jac else None
        Jp = np.zeros((self.no, 2, 3)) if jac else None
        s = self._struct()
        cost = lib().orc_ba_evaluate(C.byref(s), _p(r), _p(Jc), _p(Jp))
        return cost, r, Jc, Jp

    def normal_blocks(self, r, Jc, Jp):
        Hcc = np.zeros((self.nc, 6, 6)); gc = np.zeros((self.nc, 6))
        Hpp = np.zeros((self.np_, 3, 3)); gp = np.zeros((self.np_, 3))
        s = self._struct()
        lib().orc_ba_normal_blocks(C.byref(s), _p(r), _p(Jc), _p(Jp), _p(Hcc), _p(gc), _p(Hpp), _p(gp))
        return Hcc, gc, Hpp, gp

    def reduced_system(self, r, Jc, Jp, dc, dp, pt_begin=0, pt_end=None):
        n = 6 * self.nc
        S = np.zeros((n, n)); rhs = np.zeros(n)
        s = self._struct()
        lib().orc_ba_reduced_system(C.byref(s), _p(Jc), _p(Jp), _p(r), _p(f64(dc)), _p(f64(dp)),
                                    C.c_int(pt_begin), C.c_int(self.np_ if pt_end is None else pt_end),
                                    _p(S), _p(rhs))
        return S, rhs

    def solve(self, opt=None, **kw):
        opt = opt or default_options(**kw)
        n_rows = max(opt.max_num_iterations, opt.fixed_iterations) + 1
        trace = np.zeros((n_rows, TRACE_COLS))
        summ = LMSummary()
        s = self._struct()
        lib().orc_ba_solve(C.byref(s), C.byref(opt), C.byref(summ), _p(trace))
        return summ, trace[: summ.num_iterations + 1]

    def triangulate(self, max_iter=50):
        s = self._struct()
        lib().orc_ba_triangulate(C.byref(s), C.c_int(max_iter))


def cholesky_lower(A, threads=1):
    A = f64(A).copy()
    rc = lib().orc_cholesky_lower(_p(A), C.c_int(A.shape[0]), C.c_int(threads))
    return rc, A


def cholesky_solve(L, b):
    b = f64(b).copy()
    lib().orc_cholesky_solve(_p(f64(L)), C.c_int(L.shape[0]), _p(b))
    return b


# ---------------------------------------------------------------- dense LM with python callbacks
_RESFN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))
_PLUSFN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))


def dense_lm(residual, x0, n_res, n_local=None, plus=None, lower=None, upper=None, opt=None, **kw):
    """residual(x) -> (r[n_res], J[n_res, n_local]) ; plus(x, d) -> x_new"""
    x = f64(x0).copy()
    n_params = x.size
    n_local = n_local or n_params

    def _fn(_u, xp, rp, Jp):
        xx = np.ctypeslib.as_array(xp, shape=(n_params,))
        r, J = residual(xx.copy())
        np.ctypeslib.as_array(rp, shape=(n_res,))[:] = r
        if Jp:
            np.ctypeslib.as_array(Jp, shape=(n_res * n_local,))[:] = np.asarray(J, dtype=np.float64).reshape(-1)
        return 0

    def _plus(_u, xp, dp, op):
        xx = np.ctypeslib.as_array(xp, shape=(n_params,)).copy()
        dd = np.ctypeslib.as_array(dp, shape=(n_local,)).copy()
        np.ctypeslib.as_array(op, shape=(n_params,))[:] = plus(xx, dd)

    opt = opt or default_options(**kw)
    trace = np.zeros((opt.max_num_iterations + 1, TRACE_COLS))
    summ = LMSummary()
    cfn = _RESFN(_fn)
    cplus = _PLUSFN(_plus) if plus is not None else C.cast(None, _PLUSFN)
    lo = None if lower is None else f64(lower)
    up = None if upper is None else f64(upper)
    lib().orc_dense_lm(cfn, cplus, None, C.c_int(n_params), C.c_int(n_local), C.c_int(n_res), _p(x),
                       _p(lo), _p(up), C.byref(opt), C.byref(summ), _p(trace))
    return x, summ, trace[: summ.num_iterations + 1]


# ---------------------------------------------------------------- st17 / st7 / st3
def pnp_gauss_newton(pts_w, feats, q0, t0, rot_mode=0, max_iter=10):
    q, t = f64(q0).copy(), f64(t0).copy()
    pts_w, feats = f64(pts_w), f64(feats)
    tr = np.zeros(max_iter)
    it = lib().orc_pnp_gauss_newton(C.c_int(len(pts_w)), _p(pts_w), _p(feats), _p(q), _p(t),
                                    C.c_int(rot_mode), C.c_int(max_iter), _p(tr))
    return q, t, it, tr


def parabola_least_square(xy):
    xy = np.ascontiguousarray(xy, dtype=np.float32)
    o = np.zeros(3, dtype=np.float32)
    lib().orc_parabola_least_square(C.c_int(len(xy)), _p(xy), _p(o))
    return o


def parabola_gauss_newton(xy, iters=10):
    xy = np.ascontiguousarray(xy, dtype=np.float32)
    o = np.zeros(3, dtype=np.float32)
    it = lib().orc_parabola_gauss_newton(C.c_int(len(xy)), _p(xy), C.c_int(iters), _p(o))
    return o, it


def icp_se2_gauss_newton(pc1, pc2, iters=10):
    """st6-icp/src/include/icp.hpp:28-50 in float: returns T21 = (cos, sin, tx, ty) after `iters` iterations"""
    pc1 = np.ascontiguousarray(pc1, dtype=np.float32)
    pc2 = np.ascontiguousarray(pc2, dtype=np.float32)
    T = np.zeros(4, dtype=np.float32)
    lib().orc_icp_se2_gauss_newton(C.c_int(len(pc1)), _p(pc1), _p(pc2), C.c_int(iters), _p(T))
    return T


def calib_evaluate(params, obj, img, jac=True):
    obj, img = f64(obj), f64(img)
    V, Cn = obj.shape[0], obj.shape[1]
    e = np.zeros((V, Cn, 2))
    Ji = np.zeros((V, Cn, 2, 9)) if jac else None
    Jx = np.zeros((V, Cn, 2, 6)) if jac else None
    sse = lib().orc_calib_evaluate(C.c_int(V), C.c_int(Cn), _p(f64(params)), _p(obj), _p(img), _p(e), _p(Ji), _p(Jx))
    return sse, e, Ji, Jx


def calib_gauss_newton(params, obj, img, max_iter=10):
    params = f64(params).copy()
    obj, img = f64(obj), f64(img)
    V, Cn = obj.shape[0], obj.shape[1]
    tr = np.full(max_iter, np.nan)
    it = lib().orc_calib_gauss_newton(C.c_int(V), C.c_int(Cn), _p(params), _p(obj), _p(img), C.c_int(max_iter), _p(tr))
    return params, it, tr


# ---------------------------------------------------------------- pose graph (C4, build-defined)
class PGProblem(C.Structure):
    _fields_ = [("n_nodes", C.c_int), ("n_edges", C.c_int), ("poses", C.c_void_p), ("edge_i", C.c_void_p),
                ("edge_j", C.c_void_p), ("meas", C.c_void_p), ("node_fixed", C.c_void_p)]


class PG:
    def __init__(self, poses, edge_i, edge_j, meas, node_fixed=None):
        self.poses = f64(poses).copy().reshape(-1, 7)
        self.edge_i = np.ascontiguousarray(edge_i, dtype=np.int32)
        self.edge_j = np.ascontiguousarray(edge_j, dtype=np.int32)
        self.meas = f64(meas).reshape(-1, 7)
        self.node_fixed = None if node_fixed is None else np.ascontiguousarray(node_fixed, dtype=np.uint8)
        self.n, self.ne = len(self.poses), len(self.edge_i)

    def _struct(self):
        return PGProblem(self.n, self.ne, _p(self.poses), _p(self.edge_i), _p(self.edge_j), _p(self.meas), _p(self.node_fixed))

    def evaluate(self, jac=True):
        lib().orc_pg_evaluate.restype = C.c_double
        r = np.zeros((self.ne, 6))
        Ji = np.zeros((self.ne, 6, 6)) if jac else None
        Jj = np.zeros((self.ne, 6, 6)) if jac else None
        s = self._struct()
        cost = lib().orc_pg_evaluate(C.byref(s), _p(r), _p(Ji), _p(Jj))
        return cost, r, Ji, Jj

    def solve(self, opt=None, **kw):
        opt = opt or default_options(**kw)
        trace = np.zeros((opt.max_num_iterations + 1, TRACE_COLS))
        summ = LMSummary()
        s = self._struct()
        lib().orc_pg_solve(C.byref(s), C.byref(opt), C.byref(summ), _p(trace))
        return summ, trace[: summ.num_iterations + 1]

    def solve_sparse(self, opt=None, **kw):
        """the same LM with the normal equations solved matrix-free (certified CG): any size; returns
        (summary, trace, cg_iterations_total, worst_linear_residual)"""
        opt = opt or default_options(**kw)
        trace = np.zeros((opt.max_num_iterations + 1, TRACE_COLS))
        summ = LMSummary()
        s = self._struct()
        cg = C.c_int(); worst = C.c_double()
        lib().orc_pg_solve_sparse(C.byref(s), C.byref(opt), C.byref(summ), _p(trace), C.byref(cg), C.byref(worst))
        return summ, trace[: summ.num_iterations + 1], cg.value, worst.value


def se3_compose(a, b):
    o = np.zeros(7); lib().orc_se3_compose(_p(f64(a)), _p(f64(b)), _p(o)); return o


def se3_inverse(a):
    o = np.zeros(7); lib().orc_se3_inverse(_p(f64(a)), _p(o)); return o


def se3_retract(T, d):
    o = np.zeros(7); lib().orc_se3_retract(_p(f64(T)), _p(f64(d)), _p(o)); return o


def pg_ate(truth, est):
    lib().orc_pg_ate.restype = C.c_double
    truth, est = f64(truth), f64(est)
    return lib().orc_pg_ate(C.c_int(len(truth)), _p(truth), _p(est))
