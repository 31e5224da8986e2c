"""GPU parity tests: the HIP path (through the C ABI, slam-tricks_amd/libstba.so) against the
CPU oracle on the same seeded inputs.  FP64 everywhere; tolerances are stated per test and sit
far inside north_star's 1e-6 (residuals) / 1e-5 (pose)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def st():
    mod = importlib.import_module("slam-tricks_amd")
    assert mod.device_count() > 0, "GPU tests need a HIP device"
    return mod


def engine(st, s, **kw):
    return st.BAEngine(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"], **kw)


def oracle(O, s, **kw):
    return O.BA(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"], **kw)


def pose_err(cams, ref):
    q, qr = cams[:, :4], ref[:, :4]
    dq = np.minimum(np.abs(q - qr).max(1), np.abs(q + qr).max(1)).max()      # sign-invariant
    return dq, np.abs(cams[:, 4:] - ref[:, 4:]).max()


# ------------------------------------------------------------------------------- dense Cholesky
@pytest.mark.parametrize("n", [1, 6, 12, 127, 128, 129, 300, 777])
def test_cholesky_solve_matches_numpy(st, n):
    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, n)); A = A @ A.T + n * np.eye(n)
    b = rng.normal(size=n)
    x = st.cholesky_solve(A, b)
    assert np.allclose(x, np.linalg.solve(A, b), rtol=1e-9, atol=1e-11)
    L = st.cholesky_factor(A)
    assert np.allclose(L, np.linalg.cholesky(A), rtol=1e-10, atol=1e-10)


def test_cholesky_mfma_layout_asymmetric(st):
    """catches a transposed / mis-mapped MFMA tile: strongly asymmetric rows"""
    n = 640
    rng = np.random.default_rng(1)
    B = rng.normal(size=(n, n)) * np.linspace(0.1, 3.0, n)[:, None]
    A = B @ B.T + np.diag(np.linspace(1.0, 50.0, n))
    L = st.cholesky_factor(A)
    assert np.abs(L @ L.T - A).max() < 1e-9 * np.abs(A).max()


def test_cholesky_not_positive_definite(st):
    A = np.eye(200); A[150, 150] = -1.0
    with pytest.raises(st.StbaError) as e:
        st.cholesky_solve(A, np.ones(200))
    assert e.value.code == -4


def test_cholesky_matches_oracle(st, O):
    n = 333
    rng = np.random.default_rng(3)
    A = rng.normal(size=(n, n)); A = A @ A.T + n * np.eye(n)
    rc, Lo = O.cholesky_lower(A)
    assert rc == 0
    assert np.allclose(st.cholesky_factor(A), np.tril(Lo), rtol=1e-11, atol=1e-11)


# ------------------------------------------------------------------------------- stage kernels
@pytest.fixture(scope="module")
def small(scenes):
    return scenes.st20_scene(n_cams=12, n_pts=300, seed=5, pos_noise=0.1, ang_noise_deg=1.5, pix_noise=1e-3)


def test_residual_jacobian_elementwise(st, O, small):
    s = small
    e, o = engine(st, s), oracle(O, s)
    cost, r, Jc, Jp = e.evaluate()
    co, ro, Jco, Jpo = o.evaluate()
    assert abs(cost - co) <= 1e-13 * co
    assert np.abs(r - ro).max() < 1e-14
    assert np.abs(Jc - Jco).max() < 1e-12 and np.abs(Jp - Jpo).max() < 1e-12
    assert np.all(Jc[s["obs_cam"] == 0] == 0)            # constant camera: columns dropped
    assert abs(e.cost() - co) <= 1e-13 * co


def test_residual_jacobian_unsorted_observations(st, O, small):
    """observations in arbitrary order come back in the caller's order"""
    s = dict(small)
    rng = np.random.default_rng(0)
    p = rng.permutation(len(s["obs_cam"]))
    su = dict(s, obs_cam=s["obs_cam"][p], obs_pt=s["obs_pt"][p], obs_feat=s["obs_feat"][p])
    _, r, Jc, Jp = engine(st, su).evaluate()
    _, ro, Jco, Jpo = oracle(O, s).evaluate()
    assert np.abs(r - ro[p]).max() < 1e-14 and np.abs(Jc - Jco[p]).max() < 1e-12 and np.abs(Jp - Jpo[p]).max() < 1e-12


def test_normal_blocks(st, O, small):
    e, o = engine(st, small), oracle(O, small)
    e.evaluate()
    Hcc, gc, Hpp, gp = e.normal_blocks()
    _, ro, Jco, Jpo = o.evaluate()
    Hcco, gco, Hppo, gpo = o.normal_blocks(ro, Jco, Jpo)
    for a, b in ((Hcc, Hcco), (gc, gco), (Hpp, Hppo), (gp, gpo)):
        assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max())


def test_reduced_system_and_step(st, O, small):
    s = small
    e, o = engine(st, s), oracle(O, s)
    e.evaluate(); e.normal_blocks()
    _, ro, Jco, Jpo = o.evaluate()
    rng = np.random.default_rng(1)
    dc = rng.uniform(0.01, 0.1, (e.nc, 6)); dp = rng.uniform(0.01, 0.1, (e.np_, 3))
    S, rhs = e.reduced_system(dc, dp)
    So, rhso = o.reduced_system(ro, Jco, Jpo, dc, dp)
    scale = np.abs(So).max()
    assert np.abs(np.tril(S) - np.tril(So)).max() < 1e-11 * scale
    assert np.abs(rhs - rhso).max() < 1e-11 * max(1.0, np.abs(rhso).max())
    dxc = e.solve_reduced()
    Sf = np.tril(So) + np.tril(So, -1).T
    ref = np.linalg.solve(Sf, rhso)
    assert np.abs(dxc - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())
    dxp = e.back_substitute()
    # full-system check: (H + D) [dxc; dxp] = -g
    Hcco, gco, Hppo, gpo = o.normal_blocks(ro, Jco, Jpo)
    v = -gpo.copy()
    for i in range(o.no):
        c, j = o.obs_cam[i], o.obs_pt[i]
        v[j] -= Jpo[i].T @ (Jco[i] @ ref[6 * c:6 * c + 6])
    Hd = Hppo + np.einsum("ij,jk->ijk", dp, np.eye(3))
    ref_p = np.linalg.solve(Hd, v[..., None])[..., 0]
    assert np.abs(dxp - ref_p).max() < 1e-9 * max(1.0, np.abs(ref_p).max())
    new_cost = e.apply_step(accept=True)
    cams, pts = e.get_params()
    o2 = O.BA(cams, pts, s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
    assert abs(new_cost - o2.evaluate(jac=False)[0]) <= 1e-12 * new_cost


def test_pnp_is_ba_with_constant_landmarks(st, O, scenes, known):
    """st17 SolvePnPWith*: one camera, constant landmarks -> published truth (release.png)"""
    s = scenes.pnp_scene(seed=17)
    n = len(s["pts"])
    args = (s["pose_init"][None], s["pts"], np.zeros(n, np.int32), np.arange(n, dtype=np.int32), s["feats"])
    e = st.BAEngine(*args, pt_fixed=np.ones(n, np.uint8))
    summ, tr = e.solve()
    o = O.BA(*args, pt_fixed=np.ones(n, np.uint8))
    so, tro = o.solve()
    cams, _ = e.get_params()
    assert summ.termination_type == 0 and summ.final_cost < known["st17_pnp"]["final_cost_below"]
    assert summ.num_iterations == so.num_iterations
    dq, dt = pose_err(cams, s["pose_true"][None])
    assert dq < 1e-9 and dt < 1e-8
    dq, dt = pose_err(cams, o.cams)
    assert dq < 1e-10 and dt < 1e-10


# ------------------------------------------------------------------------------- whole solves
def test_st20_reference_size_matches_oracle_trace(st, O, scenes):
    """the reference's own BA size: 29 cameras x 600 landmarks (test_ceres.cpp:8-13)"""
    s = scenes.st20_scene()
    e, o = engine(st, s), oracle(O, s)
    summ, tr = e.solve()
    so, tro = o.solve()
    assert summ.termination_type == so.termination_type == 0
    assert summ.num_iterations == so.num_iterations
    assert summ.termination_reason == so.termination_reason
    n = min(len(tr), len(tro))
    big = tro[:n, 0] > 1e-9                      # relative agreement while the cost is above round-off
    assert np.allclose(tr[:n, 0][big], tro[:n, 0][big], rtol=1e-6)
    assert np.all(tr[:n, 6] == tro[:n, 6])       # same accept / reject sequence
    cams, pts = e.get_params()
    dq, dt = pose_err(cams, o.cams)
    assert dq < 1e-8 and dt < 1e-8               # north_star: 1e-5 on pose
    assert np.abs(pts - o.pts).max() < 1e-7
    dq, dt = pose_err(cams, s["cams_true"])
    assert dq < 1e-6 and dt < 1e-5
    assert np.all(cams[0] == s["cams0"][0]) and np.all(cams[-1] == s["cams0"][-1])


def test_noisy_problem_matched_final_cost(st, O, scenes):
    s = scenes.st20_scene(n_cams=40, n_pts=2000, max_obs_per_pt=8, seed=9, pix_noise=1e-3)
    e, o = engine(st, s), oracle(O, s)
    summ, tr = e.solve()
    so, tro = o.solve()
    assert summ.termination_type == 0
    assert abs(summ.final_cost - so.final_cost) <= 1e-6 * so.final_cost      # north_star: 1e-6 on residuals
    assert summ.num_iterations == so.num_iterations
    cams, pts = e.get_params()
    dq, dt = pose_err(cams, o.cams)
    assert dq < 1e-7 and dt < 1e-6


def test_two_view_config_c2(st, O, scenes):
    """BASELINE config C2: 2 cameras, 5k points, 10k observations"""
    s = scenes.two_view_scene(n_pts=5000)
    e, o = engine(st, s), oracle(O, s)
    summ, _ = e.solve()
    so, _ = o.solve()
    assert summ.termination_type == 0 and summ.final_cost < 1e-16
    assert summ.num_iterations == so.num_iterations
    cams, _ = e.get_params()
    dq, dt = pose_err(cams, s["cams_true"])
    assert dq < 1e-8 and dt < 1e-7
    assert cams[1, 4] == s["cams0"][1, 4]        # the gauge-fixing dof stayed put


def test_fixed_work_iterations(st, O, scenes):
    s = scenes.st20_scene(n_cams=20, n_pts=500, seed=4, pix_noise=1e-3)
    e, o = engine(st, s), oracle(O, s)
    summ, tr = e.lm_iterations(7)
    so, tro = o.solve(fixed_iterations=7)
    assert summ.num_iterations == 7 and len(tr) == 8
    assert np.allclose(tr[:, 0], tro[:, 0], rtol=1e-6)
    assert np.all(tr[:, 6] == tro[:, 6])


def test_triangulation_kernel(st, O, scenes):
    s = scenes.st20_scene(n_cams=15, n_pts=400, seed=8, retriangulate=False)
    e, o = engine(st, s), oracle(O, s)
    e.triangulate(); o.triangulate()
    _, pts = e.get_params()
    assert np.abs(pts - o.pts).max() < 1e-6      # weakly observed depths: same minimiser, different stop


def test_degenerate_inputs(st):
    """a landmark with a single observation and a camera with none"""
    cams = np.array([[0, 0, 0, 1, 0, 0, 0.0], [0, 0, 0, 1, 1, 0, 0.0], [0, 0, 0, 1, 2, 0, 0.0]])
    pts = np.array([[0.1, 0.2, 5.0], [0.5, -0.2, 6.0], [0.0, 0.0, 4.0]])
    oc = np.array([0, 1, 0, 1, 0], np.int32); op = np.array([0, 0, 1, 1, 2], np.int32)
    feat = np.array([[0.02, 0.04], [-0.18, 0.04], [0.08, -0.03], [-0.08, -0.03], [0.0, 0.0]])
    e = st.BAEngine(cams, pts, oc, op, feat, cam_fixed=np.array([[1] * 6, [0] * 6, [0] * 6], np.uint8))
    cost, r, Jc, Jp = e.evaluate()
    assert np.isfinite(cost) and r.shape == (5, 2)
    summ, _ = e.solve(max_num_iterations=5)
    assert np.isfinite(summ.final_cost)


# ------------------------------------------------------------------------------- small dense problems
def test_dense_curve_fit_c1(st, O, scenes, known):
    """BASELINE config C1: 1k residuals, 3 parameters, through the CostFunction-callback path"""
    d = scenes.curve_fit_data(1000)
    x, y = d[:, 0], d[:, 1]

    def res(p):
        return p[0] * x * x + p[1] * x + p[2] - y, np.stack([x * x, x, np.ones_like(x)], 1)
    p, summ, tr = st.dense_solve(res, [1.0, 0.0, 0.0], len(x))
    po, so, tro = O.dense_lm(res, [1.0, 0.0, 0.0], len(x))
    assert summ.termination_type == 0 and summ.num_iterations == so.num_iterations
    assert np.allclose(p, po, rtol=1e-10)
    assert np.allclose(p, [1.0, 2.0, 3.0], atol=0.1)


def test_dense_bounds_demo(st, known):
    ka = known["st17_ceres_bound"]

    def res(p):
        return np.array([p[0] - 3.0]), np.array([[1.0]])
    x, _, _ = st.dense_solve(res, [ka["x0"]], 1)
    assert abs(x[0] - ka["x_free"]) < 1e-8
    x, _, _ = st.dense_solve(res, [ka["x0"]], 1, lower=[ka["lower"]], upper=[ka["upper"]])
    assert abs(x[0] - ka["x_bounded"]) < 1e-12


def test_dense_pnp_with_manifold_callbacks(st, O, scenes):
    """SolvePnPWithSizedCostFunction-style: user Evaluate + LocalParameterization::Plus on the host"""
    s = scenes.pnp_scene(seed=17)
    n = len(s["pts"])

    def res(x):
        r = np.zeros(2 * n); J = np.zeros((2 * n, 6))
        for k in range(n):
            r[2 * k:2 * k + 2] = O.reproj_residual(x[:4], x[4:], s["pts"][k], s["feats"][k])
            J[2 * k:2 * k + 2] = O.reproj_jacobian(x[:4], x[4:], s["pts"][k], 0)[0]
        return r, J

    def plus(x, d):
        return np.concatenate([O.so3_plus(x[:4], d[:3]), x[4:] + d[3:]])
    x, summ, _ = st.dense_solve(res, s["pose_init"], 2 * n, n_local=6, plus=plus)
    assert summ.termination_type == 0 and summ.final_cost < 1e-17
    dq, dt = pose_err(x[None], s["pose_true"][None])
    assert dq < 1e-9 and dt < 1e-8


# ------------------------------------------------------------------------------- st3 calibration (a8, C3)
def test_calibration_kernel_matches_oracle(st, O, scenes):
    s = scenes.calib_scene(n_views=20, rows=8, cols=11, seed=3)            # BASELINE config C3: 20 x 88
    params = np.concatenate([s["intr_true"] * (1 + 1e-3), s["xis_true"].reshape(-1) + 1e-3])
    sse, e, Ji, Jx = st.calib_evaluate(params, s["obj"], s["img"])
    sso, eo, Jio, Jxo = O.calib_evaluate(params, s["obj"], s["img"])
    assert abs(sse - sso) <= 1e-12 * sso
    assert np.abs(e - eo).max() < 1e-9                                      # pixels, values ~1e3
    assert np.abs(Ji - Jio).max() <= 1e-11 * np.abs(Jio).max()
    assert np.abs(Jx - Jxo).max() <= 1e-11 * np.abs(Jxo).max()


def test_calibration_gauss_newton_c3(st, O, scenes):
    s = scenes.calib_scene(n_views=20, rows=8, cols=11, seed=3)
    p0 = np.concatenate([s["intr_true"][:4] * (1 + 5e-3), np.zeros(5), s["xis_true"].reshape(-1)])
    p, it, tr = st.calib_gauss_newton(p0, s["obj"], s["img"], 10)
    po, ito, tro = O.calib_gauss_newton(p0, s["obj"], s["img"], 10)
    assert it == ito
    n = np.count_nonzero(~np.isnan(tro))
    assert np.allclose(tr[:n], tro[:n], rtol=1e-9)
    assert np.allclose(p[:4], po[:4], rtol=1e-10) and np.allclose(p[4:9], po[4:9], rtol=1e-7, atol=1e-10)
    assert np.allclose(p[9:], po[9:], rtol=1e-8, atol=1e-10)
    assert np.allclose(p[:4], s["intr_true"][:4], rtol=2e-3)              # recovers the generating intrinsics


def test_calibration_real_fixture(st, O, known):
    """st3-calibration/calib/1..9.txt through the device path: recorded cost trace and intrinsics"""
    import os
    import zhang_init as Z
    from conftest import GOLDEN
    ka = known["st3_calibration"]
    obj, img = Z.read_corners(os.path.join(GOLDEN, "st3_calib"), ka["board_square_m"])
    p0 = Z.zhang_init(obj, img, lambda R, t: O.se3_log(O.rot_to_quat(R), t))
    p, it, tr = st.calib_gauss_newton(p0, obj, img, 10)
    done = tr[~np.isnan(tr)]
    assert abs(done[0] - ka["sse_first"]) < 5e-4 and abs(done[-1] - ka["sse_last"]) < 5e-4
    assert np.allclose(p[:4], ka["final_intr_dist"][:4], rtol=0, atol=6e-4)
    assert np.allclose(p[4:7], ka["final_intr_dist"][4:7], rtol=2e-6)
    assert abs(it - ka["gn_iterations"]) <= 1


@pytest.mark.gpu
def test_calibration_pipeline_from_corner_files(st, known):
    """C3 end to end through the product only: corner files (stba_corners_read) -> Zhang's closed form
    (stba_zhang_init) -> Gauss-Newton on the device (stba_calib_gauss_newton) -> the recorded answers"""
    import os
    from conftest import GOLDEN
    ka = known["st3_calibration"]
    d = os.path.join(GOLDEN, "st3_calib")
    img = []
    for f in sorted(x for x in os.listdir(d) if x.endswith(".txt")):
        rows, cols, xy = st.corners_read(os.path.join(d, f))
        img.append(xy.reshape(-1, 2))
    img = np.array(img)
    jj, ii = np.meshgrid(np.arange(cols), np.arange(rows))
    board = np.stack([jj.reshape(-1), ii.reshape(-1)], 1) * ka["board_square_m"]       # calib.cpp:28
    obj = np.repeat(board[None], len(img), 0)
    p0, _ = st.zhang_init(obj, img)
    p, it, tr = st.calib_gauss_newton(p0, obj, img, 10)
    done = tr[~np.isnan(tr)]
    assert abs(done[0] - ka["sse_first"]) < 5e-4 and abs(done[-1] - ka["sse_last"]) < 5e-4
    assert np.allclose(p[:4], ka["final_intr_dist"][:4], rtol=0, atol=6e-4)
    assert abs(it - ka["gn_iterations"]) <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1000, 1500, 2000, 3000, 4100, 6000])
def test_cholesky_persistent_kernel_is_deterministic(st, n):
    """(sizes: 8 .. 47 block columns -- every mix of the program's task forms: half panel solves in the last 30 panels, the TU
    hand-off as ten block tasks in the last 15 and as four quarter tasks before, quarter updates in the last 17)
    The persistent (dataflow) factorisation fixes the order of every floating-point operation, so
    repeated factorisations must agree bit for bit; a difference is a synchronisation or cache-coherence
    race between workgroups (this test found three).  The matrix is a low-rank product plus a small
    ridge: every trailing update matters, a stale tile gives a negative pivot."""
    rng = np.random.default_rng(5)
    B = rng.standard_normal((n, n // 2))
    A = B @ B.T + n * 0.01 * np.eye(n)
    Lref = np.linalg.cholesky(A)
    L0 = st.cholesky_factor(A)
    assert np.abs(L0 - Lref).max() <= 1e-12 * np.abs(Lref).max()
    for _ in range(7):
        assert np.array_equal(st.cholesky_factor(A), L0)
    # the solve too (the backward substitution adds its partial sums in a fixed order, no atomics): every
    # rank of a sharded run factors the same reduced system and must get the same camera step
    rhs = rng.standard_normal(n)
    x0 = st.cholesky_solve(A, rhs)
    assert np.abs(A @ x0 - rhs).max() <= 1e-9 * np.abs(rhs).max() * np.linalg.cond(A)
    for _ in range(3):
        assert np.array_equal(st.cholesky_solve(A, rhs), x0)


def test_rejected_steps_follow_the_oracle(st, O, scenes):
    """A badly initialised scene: the first three steps are REJECTED (the trust region shrinks), then the solve
    recovers.  The device loop linearises speculatively at the trial point while the host decides; a rejected step
    must put the linearisation of the old point back, in the free-running solve and in the fixed-work variant."""
    s = scenes.st20_scene(n_cams=12, n_pts=150, seed=3, pos_noise=0.6, ang_noise_deg=8.0, pix_noise=1e-3)
    o = oracle(O, s)
    so, tro = o.solve(max_num_iterations=12)
    flags = tro[: so.num_iterations + 1, 6]
    assert (flags[1:4] == 0).all() and flags[4:].all()                       # the situation this test is about
    e = engine(st, s)
    summ, tr = e.solve(max_num_iterations=12)
    assert summ.num_iterations == so.num_iterations
    assert np.array_equal(tr[: summ.num_iterations + 1, 6], flags)
    assert np.allclose(tr[: summ.num_iterations + 1, 0], tro[: so.num_iterations + 1, 0], rtol=1e-6)
    assert np.allclose(tr[: summ.num_iterations + 1, 5], tro[: so.num_iterations + 1, 5], rtol=1e-9)     # radius
    dq, dt = pose_err(e.get_params()[0], o.cams)
    assert dq < 1e-6 and dt < 1e-6
    # fixed-work variant (bench.py's): rejected steps re-linearise at the unchanged point
    e2, o2 = engine(st, s), oracle(O, s)
    summ2, tr2 = e2.lm_iterations(8)
    so2, tro2 = o2.solve(fixed_iterations=8)
    assert np.array_equal(tr2[:, 6], tro2[:, 6]) and (tr2[1:4, 6] == 0).all()
    assert np.allclose(tr2[:, 0], tro2[:, 0], rtol=1e-6)
    dq, dt = pose_err(e2.get_params()[0], o2.cams)
    assert dq < 1e-6 and dt < 1e-6


# ------------------------------------------------------------------------------- full size and edge cases
def test_c5_full_size_matches_oracle(st, O, scenes):
    """BASELINE config C5 at FULL size (1 000 cameras, 100 000 landmarks, 1 000 000 observations, the
    bench.py workload): three LM iterations of the device path against the oracle -- cost trace to 1e-6
    relative (north_star tolerance on residuals), camera poses to 1e-5."""
    s = scenes.st20_scene(n_cams=1000, n_pts=100000, max_obs_per_pt=10, seed=20, pix_noise=1e-3)
    e, o = engine(st, s), oracle(O, s)
    summ, tr = e.lm_iterations(3)
    so, tro = o.solve(fixed_iterations=3, num_threads=16)
    assert tr.shape == tro.shape
    assert np.allclose(tr[:, 0], tro[:, 0], rtol=1e-6)            # cost after every iteration
    assert np.array_equal(tr[:, 6], tro[:, 6])                    # same accept / reject decisions
    cams, pts = e.get_params()
    dq, dt = pose_err(cams, o.cams)
    assert dq < 1e-5 and dt < 1e-5
    assert np.abs(pts - o.pts).max() < 1e-5 * max(1.0, np.abs(o.pts).max())


def test_c5_converges_like_the_oracle(st, O, scenes):
    """north_star: "the same CONVERGED parameters ... within 1e-6 relative on residuals and 1e-5 on pose".  BASELINE
    config C5 at full size run to TERMINATION on both sides (Ceres' default stopping rules, test_ceres.h:148): same
    iteration count, same accept / reject sequence, same termination reason, final cost to 1e-6 relative, camera
    poses and landmarks to 1e-5.  The oracle factors its reduced system with LAPACK here (seconds per solve instead
    of minutes); its own blocked C Cholesky is the one pinned by the three-iteration test above."""
    s = scenes.st20_scene(n_cams=1000, n_pts=100000, max_obs_per_pt=10, seed=20, pix_noise=1e-3)
    e, o = engine(st, s), oracle(O, s)
    blas = O.use_lapack(True, threads=16)
    try:
        so, tro = o.solve(num_threads=16)
    finally:
        O.use_lapack(False)
    summ, tr = e.solve()
    n = so.num_iterations
    assert so.termination_type == 0 and n >= 5, (so.as_dict(), blas)
    assert summ.termination_type == 0 and summ.num_iterations == n, (summ.as_dict(), so.as_dict())
    assert summ.termination_reason == so.termination_reason
    assert np.array_equal(tr[: n + 1, 6], tro[: n + 1, 6])                 # accept / reject sequence
    assert np.allclose(tr[: n + 1, 0], tro[: n + 1, 0], rtol=1e-6)          # cost after every iteration
    assert abs(summ.final_cost - so.final_cost) <= 1e-6 * so.final_cost
    cams, pts = e.get_params()
    dq, dt = pose_err(cams, o.cams)
    assert dq < 1e-5 and dt < 1e-5, (dq, dt)
    assert np.abs(pts - o.pts).max() < 1e-5 * max(1.0, np.abs(o.pts).max())
    print(f"C5 to convergence: {n} iterations, final cost {summ.final_cost:.9e} (oracle {so.final_cost:.9e}), dq {dq:.2e} dt {dt:.2e}")


def test_edge_cases(st, O):
    """shapes the reference's data model allows (sim_data.h:38-63): everything constant, a camera nobody
    observes through, landmarks seen once, the smallest problem"""
    rng = np.random.default_rng(2)
    cams = np.zeros((3, 7)); cams[:, 3] = 1.0; cams[:, 4] = [0.0, 1.0, 2.0]
    pts = rng.uniform([-1, -1, 4], [1, 1, 6], (6, 3))
    oc = np.array([0, 1, 0, 1, 0, 1, 0, 1], np.int32)             # camera 2 observes nothing; points 4, 5 unobserved
    op = np.array([0, 0, 1, 1, 2, 2, 3, 3], np.int32)
    feat = np.array([(pts[j] - cams[c, 4:])[:2] / (pts[j] - cams[c, 4:])[2] for c, j in zip(oc, op)])
    # (1) every camera constant: only the observed landmarks move, and they are already at the optimum
    e = st.BAEngine(cams, pts + 0.01, oc, op, feat, cam_fixed=np.ones((3, 6), np.uint8))
    summ, _ = e.solve()
    c2, p2 = e.get_params()
    assert summ.termination_type == 0 and summ.final_cost < 1e-15
    assert np.array_equal(c2, cams) and np.allclose(p2[:4], pts[:4], atol=1e-6)
    assert np.array_equal(p2[4:], pts[4:] + 0.01)                 # unobserved landmarks are left alone
    # (2) everything constant: nothing to do, nothing changes, no failure
    e = st.BAEngine(cams, pts, oc, op, feat, cam_fixed=np.ones((3, 6), np.uint8), pt_fixed=np.ones(6, np.uint8))
    summ, _ = e.solve()
    c2, p2 = e.get_params()
    assert summ.termination_type == 0 and np.array_equal(c2, cams) and np.array_equal(p2, pts)
    # (3) one camera, one landmark, one observation (under-determined: the damping keeps it solvable)
    e = st.BAEngine(cams[:1], pts[:1], oc[:1] * 0, op[:1] * 0, feat[:1] + 0.01, cam_fixed=np.ones((1, 6), np.uint8))
    summ, _ = e.solve()
    assert np.isfinite(summ.final_cost) and summ.final_cost < summ.initial_cost
    # (4) same inputs through the oracle give the same iteration count and cost (case 1)
    o = O.BA(cams, pts + 0.01, oc, op, feat, np.ones((3, 6), np.uint8))
    so, _ = o.solve()
    e = st.BAEngine(cams, pts + 0.01, oc, op, feat, cam_fixed=np.ones((3, 6), np.uint8))
    summ, _ = e.solve()
    assert summ.num_iterations == so.num_iterations and abs(summ.final_cost - so.final_cost) < 1e-15


def test_landmarks_with_many_observations(st, O, scenes):
    """The landmark kernels (blocks, back-substitution) take a workgroup's 32 landmarks' records through LDS 256 at a time:
    with ~25 observations per landmark a workgroup makes several passes, and 300 landmarks leave a ragged last workgroup."""
    s = scenes.st20_scene(n_cams=120, n_pts=300, seed=5)
    per_pt = np.bincount(s["obs_pt"], minlength=300)
    assert per_pt.max() > 24 and per_pt[:288].reshape(9, 32).sum(1).max() > 3 * 256      # (four passes)
    e, o = engine(st, s), oracle(O, s)
    e.evaluate()
    Hcc, gc, Hpp, gp = e.normal_blocks()
    _, ro, Jco, Jpo = o.evaluate()
    Hcco, gco, Hppo, gpo = o.normal_blocks(ro, Jco, Jpo)
    for a, b in ((Hcc, Hcco), (gc, gco), (Hpp, Hppo), (gp, gpo)):
        assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max())
    rng = np.random.default_rng(1)
    dc = rng.uniform(0.01, 0.1, (e.nc, 6)); dp = rng.uniform(0.01, 0.1, (e.np_, 3))
    e.reduced_system(dc, dp)
    So, rhso = o.reduced_system(ro, Jco, Jpo, dc, dp)
    dxc = e.solve_reduced()
    dxp = e.back_substitute()
    v = -gpo.copy()
    for i in range(o.no):
        c, j = o.obs_cam[i], o.obs_pt[i]
        v[j] -= Jpo[i].T @ (Jco[i] @ dxc[6 * c:6 * c + 6])
    ref_p = np.linalg.solve(Hppo + np.einsum("ij,jk->ijk", dp, np.eye(3)), v[..., None])[..., 0]
    assert np.abs(dxp - ref_p).max() < 1e-10 * max(1.0, np.abs(ref_p).max())
    # and the LM loop (its back-substitution makes the trial point on the way) follows the oracle
    e, o = engine(st, s), oracle(O, s)
    summ, tr = e.solve()
    so, tro = o.solve()
    assert summ.num_iterations == so.num_iterations and summ.termination_reason == so.termination_reason
    n = min(len(tr), len(tro))
    big = tro[:n, 0] > 1e-9
    assert np.allclose(tr[:n, 0][big], tro[:n, 0][big], rtol=1e-6) and np.all(tr[:n, 6] == tro[:n, 6])
    assert np.allclose(tr[:n, 3], tro[:n, 3], rtol=1e-6, atol=1e-12)        # step norms (the fused update's statistics)
    cams, pts = e.get_params()
    dq, dt = pose_err(cams, o.cams)
    assert dq < 1e-8 and dt < 1e-8 and np.abs(pts - o.pts).max() < 1e-7


# ------------------------------------------------------------------------------- persistent program: time-out and fallback
def test_cholesky_falls_back_to_the_stage_kernels_on_a_timeout(st, O, scenes):
    """The persistent factorisation needs its workgroups resident at the same time; on a device it does not own a
    dependency may never arrive.  A workgroup then gives up after a bounded wait, the host REBUILDS the matrix and the
    stage kernels (one launch per stage and panel, nothing resident) finish the job.  Provoked here with a wait bound of
    10 ns: the result is the right one, the count goes up, and the LM loop survives it too."""
    n = 1500
    rng = np.random.default_rng(11)
    A = rng.normal(size=(n, n)); A = A @ A.T + n * np.eye(n)
    b = rng.normal(size=n)
    before = st.cholesky_timeout_count()
    st.cholesky_set_timeout_us(0.01)
    try:
        x = st.cholesky_solve(A, b)
        L = st.cholesky_factor(A)
        s = scenes.st20_scene(n_cams=40, n_pts=800, seed=9, pos_noise=0.1, ang_noise_deg=1.0, pix_noise=1e-3)
        e, o = engine(st, s), oracle(O, s)
        summ, tr = e.solve()
    finally:
        st.cholesky_set_timeout_us(0.0)
    assert st.cholesky_timeout_count() > before
    assert np.allclose(x, np.linalg.solve(A, b), rtol=1e-9, atol=1e-11)
    assert np.allclose(L, np.linalg.cholesky(A), rtol=1e-10, atol=1e-10)
    so, tro = o.solve()
    assert summ.termination_type == 0 and summ.num_iterations == so.num_iterations
    assert np.allclose(tr[: so.num_iterations + 1, 0], tro[: so.num_iterations + 1, 0], rtol=1e-9)
    dq, dt = pose_err(e.get_params()[0], o.cams)
    assert dq < 1e-8 and dt < 1e-8
    # and afterwards the persistent program is used again (after the cool-down) and still right
    for _ in range(70):
        st.cholesky_solve(A[:300, :300], b[:300])
    assert np.allclose(st.cholesky_factor(A), np.linalg.cholesky(A), rtol=1e-10, atol=1e-10)


def test_two_processes_factor_on_one_gpu(st, tmp_path):
    """two PROCESSES factoring n = 3000 systems concurrently on ONE GPU both finish with the right
    factor -- whether the device time-slices their persistent kernels (no time-out) or lets them starve each other (the
    stage kernels take over)."""
    import os, subprocess, sys
    from conftest import ROOT
    code = f"""
import importlib, sys, numpy as np
sys.path.insert(0, {ROOT!r})
st = importlib.import_module("slam-tricks_amd")
n = 3000
rng = np.random.default_rng(int(sys.argv[1]))
B = rng.normal(size=(n, n)); A = B @ B.T + n * np.eye(n)
ref = np.linalg.cholesky(A)
worst = 0.0
for k in range(12):
    L = st.cholesky_factor(A)
    worst = max(worst, float(np.abs(L - ref).max() / np.abs(ref).max()))
print("RESULT", worst, st.cholesky_timeout_count())
"""
    procs = [subprocess.Popen([sys.executable, "-c", code, str(k)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in (1, 2)]
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
        line = [l for l in out.splitlines() if l.startswith("RESULT")][0].split()
        assert float(line[1]) < 1e-11, line
        print("process: relative factor error", line[1], "time-outs (stage-kernel fallbacks)", line[2])


# ------------------------------------------------------------------------------- the dense form of the Schur complement
def _dense_visibility_scene(scenes, n_cams, n_pts, seed):
    """a 143-degree field of view: every camera sees 59 % of the cube's landmarks"""
    return scenes.st20_scene(n_cams=n_cams, n_pts=n_pts, seed=seed, pos_noise=0.1, ang_noise_deg=1.5, pix_noise=1e-3, half_w=3.0, half_h=3.0)


@pytest.mark.gpu
@pytest.mark.parametrize("n_cams,n_pts", [(12, 300), (30, 2000), (50, 777)])
def test_dense_schur_form_builds_the_same_reduced_system(st, O, scenes, n_cams, n_pts):
    """S = -(Y Y^T) on the matrix cores (STBA_SCHUR_DENSE) against the pair plan and against the oracle: same S, same rhs
    (one tile, several tiles, split K; 50 cameras = 300 rows: three tile rows with a ragged last one)"""
    s = _dense_visibility_scene(scenes, n_cams, n_pts, seed=7)
    s["cam_fixed"] = s["cam_fixed"].copy(); s["cam_fixed"][3, 4] = 1           # a constant dof in the middle as well
    e, o = engine(st, s), oracle(O, s)
    assert e.schur_mode() == e.SCHUR_PAIRS                                     # (small: the engine keeps the pair plan by itself)
    e.evaluate(); e.normal_blocks()
    _, ro, Jco, Jpo = o.evaluate()
    rng = np.random.default_rng(2)
    dc = rng.uniform(0.01, 0.1, (e.nc, 6)); dp = rng.uniform(0.01, 0.1, (e.np_, 3))
    S1, rhs1 = e.reduced_system(dc, dp)
    e.set_schur_mode(e.SCHUR_DENSE)
    assert e.schur_mode() == e.SCHUR_DENSE
    e.evaluate(); e.normal_blocks()
    S2, rhs2 = e.reduced_system(dc, dp)
    So, rhso = o.reduced_system(ro, Jco, Jpo, dc, dp)
    # (the two forms round differently -- W Hinv W^T against (W R)(W R)^T -- and the diagonal blocks are differences of large
    # sums: 1e-10 of the largest entry, where the pair plan and the oracle, which share their formula, agree to 1e-11)
    scale = np.abs(So).max()
    assert np.abs(np.tril(S2) - np.tril(So)).max() < 1e-10 * scale
    assert np.abs(np.tril(S2) - np.tril(S1)).max() < 1e-10 * scale
    assert np.abs(rhs2 - rhso).max() < 1e-10 * max(1.0, np.abs(rhso).max())
    assert np.abs(rhs2 - rhs1).max() < 1e-10 * max(1.0, np.abs(rhso).max())
    S3, _ = e.reduced_system(dc, dp)                                           # no atomics anywhere in this form: bit for bit
    assert np.array_equal(np.tril(S3), np.tril(S2))
    dxc = e.solve_reduced()
    ref = np.linalg.solve(np.tril(So) + np.tril(So, -1).T, rhso)
    assert np.abs(dxc - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())


@pytest.mark.gpu
def test_dense_schur_form_solves_like_the_pair_plan_and_the_oracle(st, O, scenes):
    s = _dense_visibility_scene(scenes, 29, 600, seed=20)
    e1, e2, o = engine(st, s), engine(st, s), oracle(O, s)
    e2.set_schur_mode(e2.SCHUR_DENSE)
    s1, t1 = e1.solve(); s2, t2 = e2.solve(); so, to = o.solve()
    assert s2.termination_type == 0 and s2.num_iterations == s1.num_iterations == so.num_iterations
    assert np.allclose(t2[:, 0], to[:, 0], rtol=1e-8, atol=1e-12)
    c1, p1 = e1.get_params(); c2, p2 = e2.get_params()
    assert max(pose_err(c2, c1)) < 1e-9 and max(pose_err(c2, o.cams)) < 1e-8 and np.abs(p2 - p1).max() < 1e-8


@pytest.mark.gpu
def test_dense_visibility_picks_the_dense_form_by_itself(st, scenes):
    """120 cameras of which 71 see any one of 8000 landmarks: 20.6 M observation pairs at 59 % visibility -- the engine takes the
    matrix-core form without being told (and the pair plan is not even built)"""
    s = _dense_visibility_scene(scenes, 120, 8000, seed=3)
    assert len(s["obs_cam"]) > 0.55 * 120 * len(s["pts0"])
    e = engine(st, s)
    assert e.schur_mode() == e.SCHUR_DENSE
    with pytest.raises(st.StbaError):
        e.set_schur_mode(e.SCHUR_PAIRS)
    c0 = e.cost()
    summ, tr = e.solve()
    assert summ.termination_type == 0 and summ.final_cost < 1e-3 * c0
    dq, dt = pose_err(e.get_params()[0], s["cams_true"])
    assert dq < 1e-3 and dt < 1e-2                                             # (1e-3 pixel noise: the truth up to the noise)


@pytest.mark.gpu
def test_dense_schur_form_with_repeated_camera_landmark_pairs(st, O, scenes):
    """two (three) observations of one (camera, landmark) pair -- stereo residuals on one pose block: the dense form writes ONE block
    of Y per pair and must sum them (it used to keep the last writer's); against the pair plan and the oracle"""
    s = _dense_visibility_scene(scenes, 12, 300, seed=11)
    rng = np.random.default_rng(5)
    no = len(s["obs_cam"])
    twice = rng.choice(no, 200, replace=False)
    thrice = twice[:40]
    extra = np.concatenate([twice, thrice])
    s["obs_cam"] = np.concatenate([s["obs_cam"], s["obs_cam"][extra]])
    s["obs_pt"] = np.concatenate([s["obs_pt"], s["obs_pt"][extra]])
    s["obs_feat"] = np.concatenate([s["obs_feat"], s["obs_feat"][extra] + rng.normal(0, 2e-3, (len(extra), 2))])
    order = np.argsort(s["obs_pt"], kind="stable")                           # (the oracle wants landmark-major observations)
    for k in ("obs_cam", "obs_pt", "obs_feat"):
        s[k] = s[k][order]
    e, o = engine(st, s), oracle(O, s)
    e.evaluate(); e.normal_blocks()
    _, ro, Jco, Jpo = o.evaluate()
    dc = rng.uniform(0.01, 0.1, (e.nc, 6)); dp = rng.uniform(0.01, 0.1, (e.np_, 3))
    S1, rhs1 = e.reduced_system(dc, dp)
    So, rhso = o.reduced_system(ro, Jco, Jpo, dc, dp)
    scale = np.abs(So).max()
    assert np.abs(np.tril(S1) - np.tril(So)).max() < 1e-10 * scale          # the pair plan treats them like any other pair
    e.set_schur_mode(e.SCHUR_DENSE)
    e.evaluate(); e.normal_blocks()
    S2, rhs2 = e.reduced_system(dc, dp)
    assert np.abs(np.tril(S2) - np.tril(So)).max() < 1e-10 * scale
    assert np.abs(rhs2 - rhso).max() < 1e-10 * max(1.0, np.abs(rhso).max())
    S3, _ = e.reduced_system(dc, dp)
    assert np.array_equal(np.tril(S3), np.tril(S2))                          # still no race: one writer per block of Y
    # and the whole solve, dense form against the oracle
    e2 = engine(st, s); e2.set_schur_mode(e2.SCHUR_DENSE)
    s2, t2 = e2.solve(); so, to = o.solve()
    assert s2.termination_type == 0 and s2.num_iterations == so.num_iterations
    assert np.allclose(t2[:, 0], to[:, 0], rtol=1e-8, atol=1e-12)


@pytest.mark.gpu
def test_normal_blocks_with_a_host_lineariser(st, O, small):
    """stba_ba_normal_blocks with host-linearised factors: Hcc / gc must come from the caller's camera Jacobians, not
    from the built-in closed form.  The factor here is the reprojection scaled by 2 (so 4 Hcc, 2 gc of the built-in one)."""
    e0, o = engine(st, small), oracle(O, small)
    _, r0, Jc0, Jp0 = e0.evaluate()
    H0, g0, P0, q0 = e0.normal_blocks()
    e1 = engine(st, small)
    calls = []

    def lin(cams, pts, want_jac):
        calls.append(want_jac)
        return 2.0 * r0, (2.0 * Jc0 if want_jac else None), (2.0 * Jp0 if want_jac else None)
    e1.set_host_linearizer(lin)
    c1, r1, _, _ = e1.evaluate(jac=False)
    assert calls and np.allclose(r1, 2.0 * r0, rtol=0, atol=0)
    H1, g1, P1, q1 = e1.normal_blocks()
    assert np.abs(H1 - 4.0 * H0).max() < 1e-12 * np.abs(H0).max() and np.abs(g1 - 4.0 * g0).max() < 1e-12 * np.abs(g0).max()
    assert np.abs(P1 - 4.0 * P0).max() < 1e-12 * np.abs(P0).max() and np.abs(q1 - 4.0 * q0).max() < 1e-12 * np.abs(q0).max()
    Ho, go, _, _ = o.normal_blocks(*o.evaluate()[1:])
    assert np.abs(H1 - 4.0 * Ho).max() < 1e-11 * np.abs(Ho).max()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["c5", "c2", "st20"])
def test_pair_plan_reduced_system_is_bitwise_reproducible(st, scenes, cfg):
    """north_star's "wavefront-segmented reductions" / SURVEY 4(iii): the Schur complement by the PAIR plan (the default form) is
    bit-identical from call to call -- every 6 x 6 block is accumulated by one wave of its task, in program order (round 5; until
    round 4 the eight waves of a task raced for the blocks with LDS atomics and S differed in its last bits from run to run).
    c5: BASELINE config 5 at full size; c2: two cameras with 5000 common landmarks (one block holds every pair: it is cut into
    parts over the eight waves); st20: the reference's 29 x 600.  Whole LM runs then end at ONE final cost."""
    if cfg == "c5":
        s = scenes.st20_scene(n_cams=1000, n_pts=100000, max_obs_per_pt=10, seed=20, pix_noise=1e-3)
    elif cfg == "c2":
        s = scenes.two_view_scene(n_pts=5000, seed=3)
    else:
        s = scenes.st20_scene()
    e = engine(st, s)
    assert e.schur_mode() == e.SCHUR_PAIRS
    e.evaluate(); e.normal_blocks()
    rng = np.random.default_rng(1)
    dc = rng.uniform(0.01, 0.1, (e.nc, 6)); dp = rng.uniform(0.01, 0.1, (e.np_, 3))
    S0, rhs0 = e.reduced_system(dc, dp)
    L0 = np.tril(S0)
    assert np.isfinite(L0).all() and np.abs(L0).max() > 0
    for _ in range(9 if cfg != "c5" else 5):
        S1, rhs1 = e.reduced_system(dc, dp)
        assert np.array_equal(np.tril(S1), L0) and np.array_equal(rhs1, rhs0)
    # a second engine on the same data builds the same plan and the same bits
    e2 = engine(st, s)
    e2.evaluate(); e2.normal_blocks()
    S2, rhs2 = e2.reduced_system(dc, dp)
    assert np.array_equal(np.tril(S2), L0) and np.array_equal(rhs2, rhs0)
    # whole runs: the same trace, bit for bit
    finals = set()
    for _ in range(3):
        e.set_params(s["cams0"], s["pts0"])
        summ, tr = e.lm_iterations(12)
        finals.add(float(summ.final_cost).hex())
    assert len(finals) == 1, finals


@pytest.mark.gpu
def test_dense_schur_form_against_the_oracle_where_the_tiling_is_live(st, O, scenes):
    """the matrix-core form of the Schur complement against orc_ba_reduced_system at 1260 rows: ten tile rows with a ragged last
    one (1260 = 9 x 128 + 108), 55 tiles cut into 64 K slices of nine 16-column rounds each, a constant dof in the middle of the
    matrix and a camera that sees nothing (until now the oracle comparison stopped at 300 rows)"""
    s = _dense_visibility_scene(scenes, 210, 3000, seed=9)
    s["cam_fixed"] = s["cam_fixed"].copy(); s["cam_fixed"][100, 2] = 1; s["cam_fixed"][177, 5] = 1
    dead = 63
    m = s["obs_cam"] != dead
    for k in ("obs_cam", "obs_pt", "obs_feat"):
        s[k] = s[k][m]
    e, o = engine(st, s), oracle(O, s)
    e.set_schur_mode(e.SCHUR_DENSE)
    e.evaluate(); e.normal_blocks()
    _, ro, Jco, Jpo = o.evaluate()
    rng = np.random.default_rng(4)
    dc = rng.uniform(0.01, 0.1, (e.nc, 6)); dp = rng.uniform(0.01, 0.1, (e.np_, 3))
    S2, rhs2 = e.reduced_system(dc, dp)
    So, rhso = o.reduced_system(ro, Jco, Jpo, dc, dp)
    scale = np.abs(So).max()
    assert S2.shape == (1260, 1260)
    # block by block, so that a tile that is wrong somewhere shows up as itself: (tile row, tile column) of the worst entry
    D = np.abs(np.tril(S2) - np.tril(So))
    worst = np.unravel_index(np.argmax(D), D.shape)
    assert D.max() < 1e-10 * scale, (D.max() / scale, worst[0] // 128, worst[1] // 128)
    assert np.abs(rhs2 - rhso).max() < 1e-10 * max(1.0, np.abs(rhso).max())
    blk = slice(6 * dead, 6 * dead + 6)                                     # the camera without observations: its damping only
    assert np.abs(np.tril(S2)[blk, :6 * dead]).max() == 0.0
    dxc = e.solve_reduced()
    ref = np.linalg.solve(np.tril(So) + np.tril(So, -1).T, rhso)
    assert np.abs(dxc - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())


@pytest.mark.gpu
def test_auto_selected_dense_form_follows_the_oracle_trace(st, O, scenes):
    """the engine's own choice of the dense form (120 cameras x 8000 landmarks at 59 % visibility, no pair plan) against the oracle's
    LM run: same iteration count, same accept / reject sequence, cost trace to 1e-8, poses to 1e-7 (this case
    was only held against the truth up to the noise)"""
    s = _dense_visibility_scene(scenes, 120, 8000, seed=3)
    e, o = engine(st, s), oracle(O, s)
    assert e.schur_mode() == e.SCHUR_DENSE
    so, tro = o.solve(num_threads=16)
    summ, tr = e.solve()
    n = so.num_iterations
    assert summ.termination_type == 0 and summ.num_iterations == n, (summ.as_dict(), so.as_dict())
    assert np.array_equal(tr[: n + 1, 6], tro[: n + 1, 6])
    assert np.allclose(tr[: n + 1, 0], tro[: n + 1, 0], rtol=1e-8, atol=1e-14)
    dq, dt = pose_err(e.get_params()[0], o.cams)
    assert dq < 1e-7 and dt < 1e-7, (dq, dt)


@pytest.mark.gpu
def test_more_cameras_than_the_linearise_kernels_lds_table(st, O, scenes):
    """1700 cameras: the camera table (56 B each) no longer fits the linearise kernel's LDS next to its staging tile, so the
    <cams-in-LDS = false> instantiations run (launch_linearize, ba_kernels.hip) -- residuals, Jacobians and the cost-only pass
    against the oracle, element by element"""
    s = scenes.st20_scene(n_cams=1700, n_pts=4000, max_obs_per_pt=5, seed=12, pix_noise=1e-3, retriangulate=False)
    e, o = engine(st, s), oracle(O, s)
    cost, r, Jc, Jp = e.evaluate()
    co, ro, Jco, Jpo = o.evaluate()
    assert abs(cost - co) <= 1e-12 * co
    assert np.abs(r - ro).max() < 1e-13 and np.abs(Jc - Jco).max() < 1e-11 and np.abs(Jp - Jpo).max() < 1e-11
    assert abs(e.cost() - co) <= 1e-12 * co                                  # the cost-only instantiation


@pytest.mark.gpu
def test_function_tolerance_switch_follows_the_oracle(st, O, scenes):
    """stba_lm_options::function_tolerance_takes_step = 0 (the step on which the function tolerance fires is not taken): engine and oracle
    agree on that reading as well -- same iteration count, same trace, last step refused, parameters one step behind the default's"""
    s = scenes.st20_scene(pix_noise=1e-3)
    e1, e0 = engine(st, s), engine(st, s)
    o0 = oracle(O, s)
    s1, t1 = e1.solve()
    s0, t0 = e0.solve(function_tolerance_takes_step=0)
    so, to = o0.solve(function_tolerance_takes_step=0)
    n = so.num_iterations
    assert s0.num_iterations == n == s1.num_iterations and s0.termination_reason == so.termination_reason == 2, (s0.as_dict(), s1.as_dict(), so.as_dict(), t0[:, 0], t1[:, 0], to[:, 0])
    assert np.array_equal(t0[: n + 1, 6], to[: n + 1, 6]) and t0[n, 6] == 0 and t1[n, 6] == 1
    assert np.allclose(t0[: n + 1, 0], to[: n + 1, 0], rtol=1e-9)
    assert abs(s0.final_cost - so.final_cost) <= 1e-9 * so.final_cost and s0.final_cost >= s1.final_cost
    dq, dt = pose_err(e0.get_params()[0], o0.cams)
    assert dq < 1e-8 and dt < 1e-8


@pytest.mark.gpu
def test_few_camera_rows_cut_into_two_slices_match_the_oracle(st, O, scenes):
    """40 cameras x 100 000 landmarks (5 M observation pairs on 40 camera rows): every row's Schur task is cut into two slices (round 5:
    a task per row left most workgroup slots empty) -- reduced system against the oracle, bitwise reproducible, and the LM trace"""
    s = scenes.st20_scene(n_cams=40, n_pts=100000, max_obs_per_pt=10, seed=5, pix_noise=1e-3, retriangulate=False)
    e, o = engine(st, s), oracle(O, s)
    assert e.schur_mode() == e.SCHUR_PAIRS
    e.evaluate(); e.normal_blocks()
    _, ro, Jco, Jpo = o.evaluate()
    rng = np.random.default_rng(8)
    dc = rng.uniform(0.01, 0.1, (e.nc, 6)); dp = rng.uniform(0.01, 0.1, (e.np_, 3))
    S1, rhs1 = e.reduced_system(dc, dp)
    So, rhso = o.reduced_system(ro, Jco, Jpo, dc, dp)
    scale = np.abs(So).max()
    assert np.abs(np.tril(S1) - np.tril(So)).max() < 1e-10 * scale
    assert np.abs(rhs1 - rhso).max() < 1e-10 * max(1.0, np.abs(rhso).max())
    S2, rhs2 = e.reduced_system(dc, dp)
    assert np.array_equal(np.tril(S2), np.tril(S1)) and np.array_equal(rhs2, rhs1)
    summ, tr = e.lm_iterations(3)
    so, tro = o.solve(fixed_iterations=3, num_threads=16)
    assert np.allclose(tr[:, 0], tro[:, 0], rtol=1e-8) and np.array_equal(tr[:, 6], tro[:, 6])


def test_features_handed_over_after_creation(st, small):
    """round 6, stba_ba_set_features: an engine created with placeholder features and given the real ones afterwards (what the
    operator API does: it learns the features from the user's cost functions while a helper thread creates the engine) solves
    exactly like one created with them -- same trace, same parameters, bit for bit; observations in a shuffled order, so that the
    engine's regrouping permutation is exercised."""
    s = small
    rng = np.random.default_rng(3)
    perm = rng.permutation(len(s["obs_cam"]))
    oc, op, of = s["obs_cam"][perm], s["obs_pt"][perm], s["obs_feat"][perm]
    a = st.BAEngine(s["cams0"], s["pts0"], oc, op, of, s["cam_fixed"])
    b = st.BAEngine(s["cams0"], s["pts0"], oc, op, np.zeros_like(of), s["cam_fixed"])
    b.set_features(of)
    sa, ta = a.solve()
    sb, tb = b.solve()
    assert sa.num_iterations == sb.num_iterations and np.array_equal(ta, tb)
    ca, pa = a.get_params(); cb, pb = b.get_params()
    assert np.array_equal(ca, cb) and np.array_equal(pa, pb)
    assert st.lib().stba_ba_set_features(b._h, None) != 0          # (a null array is refused)
