"""slam-tricks_amd -- MI355X-native nonlinear-least-squares / bundle-adjustment engine.

Python here is only a thin ctypes view of the C ABI (include/stba.h) exported by
slam-tricks_amd/libstba.so (HIP kernels for gfx950) plus the seeded scene generators; the
product is the shared library.  Nothing in this package imports, links or calls oracle/.
The library has no CPU fallback: on a machine without a HIP device every compute call raises
StbaError(STBA_ERR_NO_DEVICE).

Import with  importlib.import_module("slam-tricks_amd")  (the hyphen is the repository's name).
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STBA_LIB") or os.path.join(_HERE, "libstba.so")   # STBA_LIB: kernel experiments only
TRACE_COLS = 7
TERM_REASON = {0: "none", 1: "gradient", 2: "function", 3: "parameter", 4: "max_iter",
               5: "min_radius", 6: "solver_fail", 7: "fixed", 8: "user"}
_lib = None


class StbaError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = _lib.stba_last_error().decode() if _lib is not None else ""
        super().__init__(f"{where}: status {code} ({_lib.stba_status_string(code).decode()}) {msg}")


class LMOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int), ("initial_trust_region_radius", C.c_double),
                ("max_trust_region_radius", C.c_double), ("min_trust_region_radius", C.c_double),
                ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double),
                ("max_lm_diagonal", C.c_double), ("function_tolerance", C.c_double),
                ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("jacobi_scaling", C.c_int), ("num_threads", C.c_int),
                ("minimizer_progress_to_stdout", C.c_int), ("update_state_every_iteration", C.c_int),
                ("phase_timing", C.c_int), ("function_tolerance_takes_step", C.c_int)]


class LMSummary(C.Structure):
    _fields_ = [("termination_type", C.c_int), ("termination_reason", C.c_int), ("num_iterations", C.c_int),
                ("num_successful_steps", C.c_int), ("num_unsuccessful_steps", C.c_int),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("final_radius", C.c_double),
                ("final_gradient_max_norm", C.c_double), ("seconds_total", C.c_double),
                ("ms_linearize", C.c_double), ("ms_schur", C.c_double), ("ms_solve", C.c_double),
                ("ms_backsub", C.c_double), ("ms_cost", C.c_double),
                ("ms_allreduce", C.c_double), ("allreduce_bytes", C.c_double), ("allreduce_calls", C.c_int)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["termination_reason"] = TERM_REASON.get(d["termination_reason"], "?")
        return d


ITER_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                      C.c_double, C.c_int)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
LINEARIZE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
RESIDUAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))
PLUS_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))

# every symbol include/stba.h declares (tests check the library exports all of them)
EXPORTS = ["stba_status_string", "stba_last_error", "stba_version", "stba_device_count",
           "stba_lm_default_options", "stba_ba_create", "stba_ba_destroy", "stba_ba_set_params",
           "stba_ba_get_params", "stba_ba_set_schur_mode", "stba_ba_schur_mode", "stba_ba_set_host_linearizer", "stba_ba_set_allreduce", "stba_ba_reduced_dim", "stba_ba_evaluate",
           "stba_ba_cost", "stba_ba_normal_blocks", "stba_ba_reduced_system", "stba_ba_solve_reduced",
           "stba_ba_back_substitute", "stba_ba_apply_step", "stba_ba_solve", "stba_ba_lm_iterations",
           "stba_ba_triangulate", "stba_ba_time_linearize", "stba_ba_time_schur", "stba_cholesky_factor", "stba_cholesky_solve",
           "stba_cholesky_time", "stba_cholesky_time_split", "stba_cholesky_schedule_model", "stba_cholesky_timeout_count", "stba_cholesky_set_timeout_us", "stba_cholesky_shard_model", "stba_cholesky_shard_owner", "stba_cholesky_profile", "stba_calib_evaluate", "stba_calib_gauss_newton",
           "stba_pcg_default_options", "stba_pg_create", "stba_pg_destroy", "stba_pg_set_allreduce", "stba_pg_get_poses", "stba_pg_evaluate",
           "stba_pg_solve", "stba_pg_last_pcg_summary", "stba_pg_time_kernels", "stba_dense_solve", "stba_corners_read", "stba_corners_write", "stba_zhang_init", "stba_two_view_init", "stba_odometry_read", "stba_odometry_write", "stba_trajectory_ate",
           "stba_comm_unique_id", "stba_comm_create", "stba_comm_destroy", "stba_comm_rank", "stba_comm_allreduce_sum",
           "stba_comm_allreduce_hook", "stba_ba_set_comm", "stba_pg_set_comm",
           "stba_ba_set_features", "stba_get_device", "stba_set_device"]


def lib():
    """Loads libstba.so; raises if it has not been built (no fallback of any kind)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python __graft_entry__.py build` "
                              "(hipcc, gfx950).  There is no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.  If libstba (linked
        # against /opt/rocm) were loaded first and torch imported later, two runtimes would coexist
        # and torch.cuda would report "No HIP GPUs are available".  Importing torch first makes
        # libstba bind to the runtime torch already loaded (same soname).  Opt out with
        # STBA_NO_TORCH_PRELOAD=1 (pure C/C++ hosts never see this: they link /opt/rocm directly).
        if "torch" not in sys.modules and os.environ.get("STBA_NO_TORCH_PRELOAD", "0") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        L = C.CDLL(LIB_PATH)
        L.stba_status_string.restype = C.c_char_p
        L.stba_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def _chk(code, where):
    if code != 0:
        raise StbaError(code, where)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def device_count():
    return lib().stba_device_count()


def default_options(**kw):
    o = LMOptions()
    lib().stba_lm_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


COMM_ID_BYTES = 128


def comm_unique_id():
    """rank 0: the 128-byte id every rank passes to Comm(...) (ncclGetUniqueId)"""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _chk(lib().stba_comm_unique_id(buf), "stba_comm_unique_id")
    return buf.raw


class Comm:
    """Native RCCL communicator (one per process / GPU): the engines all-reduce on their own stream through it,
    no Python on the data path."""

    def __init__(self, uid, rank, world, device=-1):
        self._h = C.c_void_p()
        self.rank, self.world = int(rank), int(world)
        _chk(lib().stba_comm_create(C.byref(self._h), C.c_char_p(bytes(uid)), self.rank, self.world, int(device)), "stba_comm_create")

    def allreduce_sum(self, ptr, count, stream=None):
        _chk(lib().stba_comm_allreduce_sum(self._h, C.c_void_p(ptr), C.c_size_t(count), C.c_void_p(stream or 0)),
             "stba_comm_allreduce_sum")

    def close(self):
        if self._h:
            lib().stba_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BAEngine:
    """Device-resident bundle-adjustment problem (one landmark shard per engine)."""

    def __init__(self, cams, pts, obs_cam, obs_pt, obs_feat, cam_fixed=None, pt_fixed=None, stream=None):
        self._h = C.c_void_p()
        cams = _f64(cams).reshape(-1, 7)
        pts = _f64(pts).reshape(-1, 3)
        oc = np.ascontiguousarray(obs_cam, dtype=np.int32)
        op = np.ascontiguousarray(obs_pt, dtype=np.int32)
        of = _f64(obs_feat).reshape(-1, 2)
        cf = None if cam_fixed is None else np.ascontiguousarray(cam_fixed, dtype=np.uint8).reshape(-1, 6)
        pf = None if pt_fixed is None else np.ascontiguousarray(pt_fixed, dtype=np.uint8)
        self.nc, self.np_, self.no = len(cams), len(pts), len(oc)
        self._keep = []
        _chk(lib().stba_ba_create(C.byref(self._h), self.nc, self.np_, self.no, _p(cams), _p(pts), _p(oc), _p(op),
                                  _p(of), _p(cf), _p(pf), C.c_void_p(stream or 0)), "stba_ba_create")

    def close(self):
        if self._h:
            lib().stba_ba_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters
    def set_params(self, cams=None, pts=None):
        _chk(lib().stba_ba_set_params(self._h, _p(None if cams is None else _f64(cams)),
                                      _p(None if pts is None else _f64(pts))), "stba_ba_set_params")

    def set_features(self, obs_feat):
        """the observations' features again (n_obs x 2, the order given at creation): stba_ba_set_features"""
        f = _f64(obs_feat)
        assert f.size == 2 * self.no
        _chk(lib().stba_ba_set_features(self._h, _p(f)), "stba_ba_set_features")

    def get_params(self):
        cams = np.zeros((self.nc, 7)); pts = np.zeros((self.np_, 3))
        _chk(lib().stba_ba_get_params(self._h, _p(cams), _p(pts)), "stba_ba_get_params")
        return cams, pts

    def set_allreduce(self, fn, rank, world):
        cb = ALLREDUCE_FN(fn)
        self._keep.append(cb)
        _chk(lib().stba_ba_set_allreduce(self._h, cb, None, rank, world), "stba_ba_set_allreduce")

    def set_host_linearizer(self, fn):
        """fn(cams[nc,7], pts[np,3], want_jac) -> (r[no,2], Jc[no,2,6] | None, Jp[no,2,3] | None) in the CALLER's observation order
        (stba_ba_set_host_linearizer: the factors are evaluated on the host, everything behind them runs on the device)"""
        nc, np_, no = self.nc, self.np_, self.no

        def tramp(user, cams_p, pts_p, r_p, jc_p, jp_p):
            try:
                cams = np.ctypeslib.as_array(C.cast(cams_p, C.POINTER(C.c_double)), (nc, 7))
                pts = np.ctypeslib.as_array(C.cast(pts_p, C.POINTER(C.c_double)), (np_, 3))
                want = bool(jc_p)
                r, Jc, Jp = fn(cams, pts, want)
                np.ctypeslib.as_array(C.cast(r_p, C.POINTER(C.c_double)), (no, 2))[:] = r
                if want:
                    np.ctypeslib.as_array(C.cast(jc_p, C.POINTER(C.c_double)), (no, 2, 6))[:] = Jc
                    np.ctypeslib.as_array(C.cast(jp_p, C.POINTER(C.c_double)), (no, 2, 3))[:] = Jp
                return 0
            except Exception:
                return 1
        cb = LINEARIZE_FN(tramp)
        self._keep.append(cb)
        _chk(lib().stba_ba_set_host_linearizer(self._h, cb, None), "stba_ba_set_host_linearizer")

    def set_comm(self, comm):
        """landmark shard of a multi-GPU solve: cross-rank sums through a native RCCL communicator"""
        self._keep.append(comm)
        _chk(lib().stba_ba_set_comm(self._h, comm._h if comm is not None else None), "stba_ba_set_comm")

    def reduced_dim(self):
        n, npad = C.c_int(), C.c_int()
        _chk(lib().stba_ba_reduced_dim(self._h, C.byref(n), C.byref(npad)), "stba_ba_reduced_dim")
        return n.value, npad.value

    # ---- stages
    def evaluate(self, jac=True, residuals=True):
        cost = C.c_double()
        r = np.zeros((self.no, 2)) if residuals else None
        Jc = np.zeros((self.no, 2, 6)) if jac else None
        Jp = np.zeros((self.no, 2, 3)) if jac else None
        _chk(lib().stba_ba_evaluate(self._h, C.byref(cost), _p(r), _p(Jc), _p(Jp)), "stba_ba_evaluate")
        return cost.value, r, Jc, Jp

    def cost(self):
        c = C.c_double()
        _chk(lib().stba_ba_cost(self._h, C.byref(c)), "stba_ba_cost")
        return c.value

    def normal_blocks(self):
        Hcc = np.zeros((self.nc, 6, 6)); gc = np.zeros((self.nc, 6))
        Hpp = np.zeros((self.np_, 3, 3)); gp = np.zeros((self.np_, 3))
        _chk(lib().stba_ba_normal_blocks(self._h, _p(Hcc), _p(gc), _p(Hpp), _p(gp)), "stba_ba_normal_blocks")
        return Hcc, gc, Hpp, gp

    def reduced_system(self, dc, dp, fetch=True):
        n = 6 * self.nc
        S = np.zeros((n, n)) if fetch else None
        rhs = np.zeros(n) if fetch else None
        _chk(lib().stba_ba_reduced_system(self._h, _p(_f64(dc)), _p(_f64(dp)), _p(S), _p(rhs)),
             "stba_ba_reduced_system")
        return S, rhs

    def solve_reduced(self):
        dxc = np.zeros(6 * self.nc)
        _chk(lib().stba_ba_solve_reduced(self._h, _p(dxc)), "stba_ba_solve_reduced")
        return dxc

    def back_substitute(self):
        dxp = np.zeros((self.np_, 3))
        _chk(lib().stba_ba_back_substitute(self._h, _p(dxp)), "stba_ba_back_substitute")
        return dxp

    def apply_step(self, accept=True):
        c = C.c_double()
        _chk(lib().stba_ba_apply_step(self._h, int(bool(accept)), C.byref(c)), "stba_ba_apply_step")
        return c.value

    # ---- solves
    def solve(self, opt=None, callback=None, **kw):
        opt = opt or default_options(**kw)
        trace = np.zeros((opt.max_num_iterations + 1, TRACE_COLS))
        summ = LMSummary()
        cb = ITER_CB(callback) if callback else C.cast(None, ITER_CB)
        _chk(lib().stba_ba_solve(self._h, C.byref(opt), C.byref(summ), _p(trace), cb, None), "stba_ba_solve")
        return summ, trace[: summ.num_iterations + 1]

    def lm_iterations(self, iterations, opt=None, **kw):
        opt = opt or default_options(**kw)
        trace = np.zeros((iterations + 1, TRACE_COLS))
        summ = LMSummary()
        _chk(lib().stba_ba_lm_iterations(self._h, C.byref(opt), int(iterations), C.byref(summ), _p(trace)),
             "stba_ba_lm_iterations")
        return summ, trace

    def triangulate(self, max_iter=50):
        _chk(lib().stba_ba_triangulate(self._h, int(max_iter)), "stba_ba_triangulate")

    def time_linearize(self, reps=20):
        ms = C.c_double()
        _chk(lib().stba_ba_time_linearize(self._h, int(reps), C.byref(ms)), "stba_ba_time_linearize")
        return ms.value

    SCHUR_AUTO, SCHUR_PAIRS, SCHUR_DENSE = 0, 1, 2

    def set_schur_mode(self, mode):
        """form of the Schur complement: SCHUR_PAIRS (pair plan, LDS accumulation) | SCHUR_DENSE (S = -(Y Y^T) on the matrix cores,
        for dense visibility) | SCHUR_AUTO (what the engine chose at creation)"""
        _chk(lib().stba_ba_set_schur_mode(self._h, int(mode)), "stba_ba_set_schur_mode")

    def schur_mode(self):
        m = C.c_int()
        _chk(lib().stba_ba_schur_mode(self._h, C.byref(m)), "stba_ba_schur_mode")
        return m.value

    def time_schur(self, reps=10):
        """(ms per launch of the Schur-complement kernel, LDS atomics per launch, observation pairs per launch)"""
        ms, at, pr = C.c_double(), C.c_double(), C.c_double()
        _chk(lib().stba_ba_time_schur(self._h, int(reps), C.byref(ms), C.byref(at), C.byref(pr)), "stba_ba_time_schur")
        return ms.value, at.value, pr.value


class PCGOptions(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("relative_tolerance", C.c_double), ("check_every", C.c_int),
                ("forcing_eta0", C.c_double), ("forcing_eta_min", C.c_double), ("coarse_group", C.c_int),
                ("coarse_refresh_every", C.c_int), ("one_kernel_solve", C.c_int),
                ("coarse_async", C.c_int), ("forcing_eta_final", C.c_double), ("coarse_eta", C.c_double), ("coarse_async_after", C.c_int),
                ("coarse_async_decrease", C.c_double)]


class PCGSummary(C.Structure):
    _fields_ = [("iterations_total", C.c_int), ("solves", C.c_int), ("hit_cap", C.c_int), ("max_iterations_in_a_solve", C.c_int),
                ("coarse_dim", C.c_int), ("coarse_refreshes", C.c_int), ("last_eta", C.c_double),
                ("coarse_failures", C.c_int), ("one_kernel_solves", C.c_int), ("linear_solve_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class PGEngine:
    """Device-resident pose graph (BASELINE config C4, build-defined)."""

    def __init__(self, poses, edge_i, edge_j, meas, node_fixed=None, stream=None):
        self._h = C.c_void_p()
        poses = _f64(poses).reshape(-1, 7)
        ei = np.ascontiguousarray(edge_i, dtype=np.int32); ej = np.ascontiguousarray(edge_j, dtype=np.int32)
        meas = _f64(meas).reshape(-1, 7)
        nf = None if node_fixed is None else np.ascontiguousarray(node_fixed, dtype=np.uint8)
        self.n, self.m = len(poses), len(ei)
        _chk(lib().stba_pg_create(C.byref(self._h), self.n, self.m, _p(poses), _p(ei), _p(ej), _p(meas), _p(nf),
                                  C.c_void_p(stream or 0)), "stba_pg_create")

    def close(self):
        if self._h:
            lib().stba_pg_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_allreduce(self, fn, rank, world):
        cb = ALLREDUCE_FN(fn)
        self._keep = getattr(self, "_keep", []) + [cb]
        _chk(lib().stba_pg_set_allreduce(self._h, cb, None, rank, world), "stba_pg_set_allreduce")

    def set_comm(self, comm):
        """edge shard of a multi-GPU solve: cross-rank sums through a native RCCL communicator"""
        self._keep = getattr(self, "_keep", []) + [comm]
        _chk(lib().stba_pg_set_comm(self._h, comm._h if comm is not None else None), "stba_pg_set_comm")

    def get_poses(self):
        out = np.zeros((self.n, 7))
        _chk(lib().stba_pg_get_poses(self._h, _p(out)), "stba_pg_get_poses")
        return out

    def evaluate(self, jac=True):
        cost = C.c_double()
        r = np.zeros((self.m, 6))
        Ji = np.zeros((self.m, 6, 6)) if jac else None
        Jj = np.zeros((self.m, 6, 6)) if jac else None
        _chk(lib().stba_pg_evaluate(self._h, C.byref(cost), _p(r), _p(Ji), _p(Jj)), "stba_pg_evaluate")
        return cost.value, r, Ji, Jj

    def time_kernels(self, reps=50):
        """(ms per residual+Jacobian launch, ms per matrix-free product), hipEvent-timed on the device"""
        a, b = C.c_double(), C.c_double()
        _chk(lib().stba_pg_time_kernels(self._h, int(reps), C.byref(a), C.byref(b)), "stba_pg_time_kernels")
        return a.value, b.value

    def pcg_options(self, **kw):
        pcg = PCGOptions()
        lib().stba_pcg_default_options(C.byref(pcg))
        for k, v in kw.items():
            setattr(pcg, k, v)
        return pcg

    def pcg_summary(self):
        """linear-solver side of the last solve(): PCG iterations, solves that hit the cap, coarse dimension, ..."""
        out = PCGSummary()
        _chk(lib().stba_pg_last_pcg_summary(self._h, C.byref(out)), "stba_pg_last_pcg_summary")
        return out

    def solve(self, opt=None, pcg=None, **kw):
        opt = opt or default_options(**kw)
        if pcg is None:
            pcg = self.pcg_options()
        trace = np.zeros((opt.max_num_iterations + 1, TRACE_COLS))
        summ = LMSummary()
        total = C.c_int()
        _chk(lib().stba_pg_solve(self._h, C.byref(opt), C.byref(pcg), C.byref(summ), _p(trace), C.byref(total)),
             "stba_pg_solve")
        return summ, trace[: summ.num_iterations + 1], total.value


def cholesky_factor(A, stream=None):
    A = _f64(A).copy()
    _chk(lib().stba_cholesky_factor(_p(A), A.shape[0], C.c_void_p(stream or 0)), "stba_cholesky_factor")
    return np.tril(A)


def cholesky_solve(A, b, stream=None):
    A = _f64(A); x = _f64(b).copy()
    _chk(lib().stba_cholesky_solve(_p(A), A.shape[0], _p(x), C.c_void_p(stream or 0)), "stba_cholesky_solve")
    return x


def cholesky_time(n, reps=5, stream=None):
    ms = C.c_double()
    _chk(lib().stba_cholesky_time(int(n), int(reps), C.byref(ms), C.c_void_p(stream or 0)), "stba_cholesky_time")
    return ms.value


def cholesky_schedule_model(n, n_xcd=8, wg_per_xcd=32):
    """Host-only: makespan (us) the scheduling model predicts for the persistent factorisation kernel."""
    out = C.c_double()
    _chk(lib().stba_cholesky_schedule_model(int(n), int(n_xcd), int(wg_per_xcd), C.byref(out)), "stba_cholesky_schedule_model")
    return out.value


def cholesky_set_timeout_us(us):
    """bound on a workgroup's wait for a dependency inside the persistent factorisation (0: automatic)"""
    _chk(lib().stba_cholesky_set_timeout_us(C.c_double(us)), "stba_cholesky_set_timeout_us")


def cholesky_timeout_count():
    """how often the persistent factorisation gave up (shared device) and the stage kernels took over, in this process"""
    return int(lib().stba_cholesky_timeout_count())


def cholesky_shard_model(n, n_gpus, n_xcd=8, wg_per_xcd=32, rows_per_group=0, hop_us=3.0, link_gb_per_s=48.0):
    """Host-only design study: (makespan us, cross-GPU dependencies, remote 128x128 tiles fetched by the busiest GPU)
    of the factorisation's task graph spread block-cyclically over n_gpus GPUs (include/stba.h)."""
    ms = C.c_double(); ce = C.c_double(); ti = C.c_double()
    _chk(lib().stba_cholesky_shard_model(int(n), int(n_gpus), int(n_xcd), int(wg_per_xcd), int(rows_per_group), C.c_double(hop_us),
                                         C.c_double(link_gb_per_s), C.byref(ms), C.byref(ce), C.byref(ti)), "stba_cholesky_shard_model")
    return ms.value, ce.value, ti.value


def cholesky_shard_owner(n_block_rows, n_gpus, rows_per_group):
    """tile row -> GPU of the block-cyclic distribution of the sharded-solve design"""
    out = np.zeros(int(n_block_rows), np.int32)
    _chk(lib().stba_cholesky_shard_owner(int(n_block_rows), int(n_gpus), int(rows_per_group), out.ctypes.data_as(C.POINTER(C.c_int))),
         "stba_cholesky_shard_owner")
    return out


def cholesky_time_split(n, reps=5, stream=None):
    """(ms_factor, ms_backward) of the production schedule, hipEvent-timed on the device"""
    a = C.c_double(); b = C.c_double()
    _chk(lib().stba_cholesky_time_split(int(n), int(reps), C.byref(a), C.byref(b), C.c_void_p(stream or 0)),
         "stba_cholesky_time_split")
    return a.value, b.value


def cholesky_profile(n, stream=None):
    ms = np.zeros(4); fl = C.c_double(); flp = C.c_double(); nl = C.c_int()
    _chk(lib().stba_cholesky_profile(int(n), _p(ms), C.byref(fl), C.byref(flp), C.byref(nl), C.c_void_p(stream or 0)),
         "stba_cholesky_profile")
    return dict(ms_diag=ms[0], ms_trsm=ms[1], ms_syrk=ms[2], ms_bwd=ms[3], syrk_flops=fl.value,
                syrk_flops_padded=flp.value, syrk_launches=nl.value)


def two_view_init(f1, f2, K, points=True, stream=None):
    """two_view_geometry.cpp:18-126 on the device: pixel pairs (n, 2) x 2 and K -> dict(F, R, t, pts, fails);
    (R, t) = pose of frame 2 in frame 1, |t| = 1.  Raises StbaError (no solution) like the reference's empty optional."""
    f1, f2, K = _f64(f1), _f64(f2), _f64(K)
    n = f1.shape[0]
    F = np.zeros((3, 3)); R = np.zeros((3, 3)); t = np.zeros(3)
    pts = np.zeros((n, 3)) if points else None
    fails = np.zeros(4, dtype=np.int32)
    _chk(lib().stba_two_view_init(n, _p(f1), _p(f2), _p(K), _p(F), _p(R), _p(t), _p(pts) if points else None,
                                  fails.ctypes.data_as(C.POINTER(C.c_int)), C.c_void_p(stream or 0)), "stba_two_view_init")
    return dict(F=F, R=R, t=t, pts=pts, fails=fails)


def odometry_read(path):
    """odometry file (st16 scene.cpp:66-110) -> (stamps[n], poses[n, 7] = qx qy qz qw x y z)"""
    n = C.c_int()
    _chk(lib().stba_odometry_read(path.encode(), C.byref(n), None, None, 0), "stba_odometry_read")
    stamps = np.zeros(n.value); poses = np.zeros((n.value, 7))
    _chk(lib().stba_odometry_read(path.encode(), C.byref(n), _p(stamps), _p(poses), n.value), "stba_odometry_read")
    return stamps, poses


def odometry_write(path, stamps, poses):
    poses = _f64(poses).reshape(-1, 7)
    st_ = None if stamps is None else _f64(stamps)
    _chk(lib().stba_odometry_write(path.encode(), len(poses), _p(st_), _p(poses)), "stba_odometry_write")


def trajectory_ate(truth, estimate):
    """absolute trajectory error (st4 pose_simulation.cpp:198-209) of two (n, 7) pose sequences"""
    a, b = _f64(truth).reshape(-1, 7), _f64(estimate).reshape(-1, 7)
    out = C.c_double()
    _chk(lib().stba_trajectory_ate(len(a), _p(a), _p(b), C.byref(out)), "stba_trajectory_ate")
    return out.value


def corners_read(path):
    """chessboard corner file -> (rows, cols, xy[rows, cols, 2])  (cbcorner.cpp:50-73)"""
    r = C.c_int(); c = C.c_int()
    _chk(lib().stba_corners_read(path.encode(), C.byref(r), C.byref(c), None, 0), "stba_corners_read")
    xy = np.zeros((r.value, c.value, 2))
    _chk(lib().stba_corners_read(path.encode(), C.byref(r), C.byref(c), _p(xy), r.value * c.value), "stba_corners_read")
    return r.value, c.value, xy


def corners_write(path, xy):
    xy = _f64(xy)
    _chk(lib().stba_corners_write(path.encode(), xy.shape[0], xy.shape[1], _p(xy)), "stba_corners_write")


def zhang_init(obj, img):
    """closed-form calibration start (calib.cpp:55-173): obj, img (V, C, 2) -> (params[9 + 6V], H[V, 3, 3])"""
    obj, img = _f64(obj), _f64(img)
    V, Cn = obj.shape[0], obj.shape[1]
    params = np.zeros(9 + 6 * V); H = np.zeros((V, 3, 3))
    _chk(lib().stba_zhang_init(V, Cn, _p(obj), _p(img), _p(params), _p(H)), "stba_zhang_init")
    return params, H


def calib_evaluate(params, obj, img, jac=True):
    """st3 calibration residual / Jacobian kernel.  obj, img: (V, C, 2)"""
    obj, img = _f64(obj), _f64(img)
    V, Cn = obj.shape[0], obj.shape[1]
    sse = C.c_double()
    e = np.zeros((V, Cn, 2))
    Ji = np.zeros((V, Cn, 2, 9)) if jac else None
    Jx = np.zeros((V, Cn, 2, 6)) if jac else None
    _chk(lib().stba_calib_evaluate(V, Cn, _p(_f64(params)), _p(obj), _p(img), C.byref(sse), _p(e), _p(Ji), _p(Jx)),
         "stba_calib_evaluate")
    return sse.value, e, Ji, Jx


def calib_gauss_newton(params, obj, img, max_iter=10):
    """CalibSolver::totalOptimization on the device; returns (params, iterations, sse_trace)"""
    params = _f64(params).copy()
    obj, img = _f64(obj), _f64(img)
    V, Cn = obj.shape[0], obj.shape[1]
    tr = np.full(max_iter, np.nan)
    it = C.c_int()
    _chk(lib().stba_calib_gauss_newton(V, Cn, _p(params), _p(obj), _p(img), int(max_iter), _p(tr), C.byref(it)),
         "stba_calib_gauss_newton")
    return params, it.value, tr


def dense_solve(residual, x0, n_res, n_local=None, plus=None, lower=None, upper=None, opt=None, callback=None, **kw):
    """residual(x) -> (r[n_res], J[n_res, n_local] or None-ignored); plus(x, d) -> x_new.
    The residual/Jacobian callback runs on the host (it is user code, like
    CostFunction::Evaluate); normal equations and the damped Cholesky step run on the device."""
    x = _f64(x0).copy()
    n_params = x.size
    n_local = n_local or n_params

    def _fn(_u, xp, rp, Jp):
        xx = np.ctypeslib.as_array(xp, shape=(n_params,)).copy()
        r, J = residual(xx)
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
    cfn = RESIDUAL_FN(_fn)
    cplus = PLUS_FN(_plus) if plus is not None else C.cast(None, PLUS_FN)
    cb = ITER_CB(callback) if callback else C.cast(None, ITER_CB)
    lo = None if lower is None else _f64(lower)
    up = None if upper is None else _f64(upper)
    _chk(lib().stba_dense_solve(cfn, cplus, None, n_params, n_local, n_res, _p(x), _p(lo), _p(up), C.byref(opt),
                                C.byref(summ), _p(trace), cb, None), "stba_dense_solve")
    return x, summ, trace[: summ.num_iterations + 1]
