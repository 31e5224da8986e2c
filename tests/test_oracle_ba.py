"""CPU tests of the oracle's bundle-adjustment path: Schur reduction vs the full normal
equations, landmark-shard additivity, LM convergence on the reference-sized st20
scene, and an independent cross-check of the fixed point with scipy.optimize.least_squares."""
import numpy as np
import pytest
from scipy.optimize import least_squares


def small_scene(scenes, **kw):
    args = dict(n_cams=6, n_pts=40, seed=7, pos_noise=0.05, ang_noise_deg=1.0)
    args.update(kw)
    return scenes.st20_scene(**args)


def make_ba(O, s, **kw):
    return O.BA(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"], **kw)


def full_normal_solution(ba, r, Jc, Jp, dc, dp):
    """dense (6C+3P) normal equations with numpy, fixed dofs pinned"""
    nc, npt = ba.nc, ba.np_
    n = 6 * nc + 3 * npt
    J = np.zeros((2 * ba.no, n))
    for i in range(ba.no):
        c, j = ba.obs_cam[i], ba.obs_pt[i]
        J[2 * i:2 * i + 2, 6 * c:6 * c + 6] = Jc[i]
        J[2 * i:2 * i + 2, 6 * nc + 3 * j:6 * nc + 3 * j + 3] = Jp[i]
    H = J.T @ J + np.diag(np.concatenate([dc.reshape(-1), dp.reshape(-1)]))
    g = J.T @ r.reshape(-1)
    fixed = np.concatenate([ba.cam_fixed.reshape(-1).astype(bool), np.zeros(3 * npt, bool)])
    H[fixed, :] = 0; H[:, fixed] = 0; H[fixed, fixed] = 1.0; g[fixed] = 0
    return np.linalg.solve(H, -g)


def test_normal_blocks_match_dense(O, scenes):
    s = small_scene(scenes)
    ba = make_ba(O, s)
    cost, r, Jc, Jp = ba.evaluate()
    assert abs(cost - 0.5 * (r ** 2).sum()) < 1e-15 * max(1, cost)
    Hcc, gc, Hpp, gp = ba.normal_blocks(r, Jc, Jp)
    for c in range(ba.nc):
        m = ba.obs_cam == c
        assert np.allclose(Hcc[c], np.einsum("nki,nkj->ij", Jc[m], Jc[m]), atol=1e-12)
        assert np.allclose(gc[c], np.einsum("nki,nk->i", Jc[m], r[m]), atol=1e-12)
    for j in range(ba.np_):
        m = ba.obs_pt == j
        assert np.allclose(Hpp[j], np.einsum("nki,nkj->ij", Jp[m], Jp[m]), atol=1e-12)
    assert np.all(Jc[ba.obs_cam == 0] == 0) and np.all(Jc[ba.obs_cam == ba.nc - 1] == 0)   # constant cameras


def test_schur_equals_full_system(O, scenes):
    s = small_scene(scenes)
    ba = make_ba(O, s)
    _, r, Jc, Jp = ba.evaluate()
    rng = np.random.default_rng(0)
    dc = rng.uniform(0.01, 0.1, (ba.nc, 6)); dp = rng.uniform(0.01, 0.1, (ba.np_, 3))
    S, rhs = ba.reduced_system(r, Jc, Jp, dc, dp)
    Sfull = np.tril(S) + np.tril(S, -1).T
    dxc = np.linalg.solve(Sfull, rhs)
    ref = full_normal_solution(ba, r, Jc, Jp, dc, dp)
    assert np.allclose(dxc, ref[:6 * ba.nc], atol=1e-9)
    rc, L = O.cholesky_lower(S)
    assert rc == 0
    assert np.allclose(O.cholesky_solve(L, rhs), dxc, atol=1e-10)


def test_landmark_shards_add_up(O, scenes):
    """per-shard partial reduced systems sum to the whole one."""
    s = small_scene(scenes, n_pts=60)
    ba = make_ba(O, s)
    _, r, Jc, Jp = ba.evaluate()
    dc = np.full((ba.nc, 6), 0.03); dp = np.full((ba.np_, 3), 0.02)
    S, rhs = ba.reduced_system(r, Jc, Jp, dc, dp)
    cuts = [0, 17, 41, ba.np_]
    Ssum = np.zeros_like(S); rsum = np.zeros_like(rhs)
    for a, b in zip(cuts[:-1], cuts[1:]):
        Sp, rp = ba.reduced_system(r, Jc, Jp, dc, dp, a, b)
        Ssum += Sp; rsum += rp
    assert np.allclose(Ssum, S, atol=1e-11) and np.allclose(rsum, rhs, atol=1e-11)


def test_st20_reference_size_converges_to_truth(O, scenes):
    """29 cameras x 600 landmarks, noise (0.3 m, 3 deg) as test_ceres.cpp:13; first/last camera
    constant (test_ceres.h:127-130).  Noise-free observations -> pose = truth."""
    s = scenes.st20_scene()
    assert len(s["cams0"]) == 29 and len(s["pts0"]) == 600
    ba = make_ba(O, s)
    summ, tr = ba.solve()
    assert summ.termination_type == 0
    assert summ.initial_cost > 1.0 and summ.final_cost < 1e-11      # float-rounded features floor
    assert summ.num_iterations <= 12
    q, qt = ba.cams[:, :4], s["cams_true"][:, :4]
    assert np.minimum(np.abs(q - qt).max(1), np.abs(q + qt).max(1)).max() < 1e-6
    assert np.abs(ba.cams[:, 4:] - s["cams_true"][:, 4:]).max() < 1e-5
    assert np.all(ba.cams[0] == s["cams0"][0]) and np.all(ba.cams[-1] == s["cams0"][-1])
    assert np.all(np.diff(tr[tr[:, 6] > 0, 0]) <= 0)                # monotone on accepted steps


def test_fixed_point_matches_scipy(O, scenes):
    """independent solver (scipy TRF) on the same noisy problem reaches the same minimiser"""
    s = small_scene(scenes, n_cams=12, n_pts=80, pix_noise=1e-3, seed=11)
    ba = make_ba(O, s)
    summ, _ = ba.solve(function_tolerance=1e-14, parameter_tolerance=1e-14, gradient_tolerance=1e-14,
                       max_num_iterations=100)
    nc, npt = ba.nc, ba.np_
    free = ~s["cam_fixed"].reshape(-1).astype(bool)
    cams0, pts0 = s["cams0"], s["pts0"]

    def unpack(x):
        d = np.zeros(6 * nc); d[free] = x[:free.sum()]
        cams = cams0.copy()
        for c in range(nc):
            cams[c, :4] = O.so3_plus(cams0[c, :4], d[6 * c:6 * c + 3])
            cams[c, 4:] = cams0[c, 4:] + d[6 * c + 3:6 * c + 6]
        return cams, pts0 + x[free.sum():].reshape(-1, 3)

    def fun(x):
        cams, pts = unpack(x)
        f, _ = scenes.project(cams, pts, s["obs_cam"], s["obs_pt"])
        return (f - s["obs_feat"]).reshape(-1)
    sol = least_squares(fun, np.zeros(free.sum() + 3 * npt), method="trf", xtol=1e-15, ftol=1e-15, gtol=1e-15)
    cams_s, pts_s = unpack(sol.x)
    assert abs(sol.cost - summ.final_cost) <= 1e-9 * summ.final_cost
    q, qs = ba.cams[:, :4], cams_s[:, :4]
    assert np.minimum(np.abs(q - qs).max(1), np.abs(q + qs).max(1)).max() < 1e-6
    assert np.abs(ba.cams[:, 4:] - cams_s[:, 4:]).max() < 1e-5
    assert np.abs(ba.pts - pts_s).max() < 1e-4


def test_triangulation_matches_scene_generator(O, scenes):
    s = small_scene(scenes, retriangulate=False)
    ba = make_ba(O, s)
    ba.triangulate()
    ref = scenes.triangulate(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"])
    assert np.abs(ba.pts - ref).max() < 1e-6


def test_two_view_scene(O, scenes):
    s = scenes.two_view_scene(n_pts=300)
    ba = make_ba(O, s)
    summ, _ = ba.solve()
    assert summ.termination_type == 0 and summ.final_cost < 1e-18
    q, qt = ba.cams[1, :4], s["cams_true"][1, :4]
    assert min(np.abs(q - qt).max(), np.abs(q + qt).max()) < 1e-8
    assert np.abs(ba.cams[1, 4:] - s["cams_true"][1, 4:]).max() < 1e-7


def test_fixed_iterations_mode(O, scenes):
    s = small_scene(scenes, pix_noise=1e-3)
    ba = make_ba(O, s)
    summ, tr = ba.solve(fixed_iterations=9)
    assert summ.num_iterations == 9 and len(tr) == 10


def test_oracle_has_not_drifted_from_its_frozen_traces(O, scenes):
    """tests/golden/oracle_traces.json (tests/golden/make_oracle_traces.py) freezes the oracle's own LM traces and
    final parameters on the reference-size scene (29 x 600) and on config C2: a change to oracle/oracle.c that moves
    them is caught here.  A self-regression fixture, NOT a reference pin (the BA leg stays 'parity unpinned')."""
    import json
    import os
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "oracle_traces.json")) as f:
        gold = json.load(f)
    for key, s in (("st20", scenes.st20_scene()), ("c2", scenes.two_view_scene(n_pts=5000))):
        g = gold[key]
        o = O.BA(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
        assert (o.nc, o.np_, o.no) == (g["n_cams"], g["n_pts"], g["n_obs"])
        summ, tr = o.solve()
        assert summ.num_iterations == g["num_iterations"] and summ.termination_type == g["termination_type"]
        assert [int(x) for x in tr[:, 6]] == g["accepted"]
        # the scenes are noise-free: the last costs are round-off (1e-13 .. 1e-21), compared absolutely
        assert np.allclose(tr[:, 0], g["cost_trace"], rtol=1e-7, atol=1e-12)
        assert np.allclose(tr[:, 5], g["radius_trace"], rtol=1e-6)
        assert np.abs(o.cams.reshape(-1) - np.array(g["final_cams"])).max() < 1e-9
        assert np.abs(o.pts[:20].reshape(-1) - np.array(g["final_pts_head"])).max() < 1e-8


def test_lapack_backed_reduced_solve_follows_the_c_factorisation(O, scenes):
    """bench.py's cpu_baseline leg may factor the reduced camera system with LAPACK (oracle_py.use_lapack): same
    iterations, same decisions, same answer as the blocked C factorisation"""
    s = scenes.st20_scene(n_cams=40, n_pts=800, max_obs_per_pt=6, seed=9, pix_noise=1e-3)
    def run():
        ba = O.BA(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
        so, tr = ba.solve(fixed_iterations=4)
        return so, tr, ba.cams.copy(), ba.pts.copy()
    so0, tr0, c0, p0 = run()
    info = O.use_lapack(True, threads=2)
    if info is None:
        pytest.skip("scipy LAPACK not importable")
    try:
        so1, tr1, c1, p1 = run()
    finally:
        O.use_lapack(False)
    assert np.array_equal(tr0[:, 6], tr1[:, 6])
    assert np.allclose(tr0[:, 0], tr1[:, 0], rtol=1e-10)
    assert np.abs(c0 - c1).max() < 1e-10 and np.abs(p0 - p1).max() < 1e-9


def test_function_tolerance_switch(O, scenes):
    """the one reading of Ceres that changes results: is the step on which the function tolerance fires taken?
    1 (default): yes, if it decreases the cost; 0 (Ceres >= 1.12's order of calls): no.  Same iterations, same trace up to the last row, one accepted step
    and one tiny cost change apart; the final parameters agree far inside north_star's tolerances either way."""
    s = scenes.st20_scene(pix_noise=1e-3)
    mk = lambda: O.BA(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
    o1, o0 = mk(), mk()
    s1, t1 = o1.solve()
    s0, t0 = o0.solve(function_tolerance_takes_step=0)
    assert O.default_options().function_tolerance_takes_step == 1
    assert s1.termination_reason == O.TERM_FUNCTION if hasattr(O, "TERM_FUNCTION") else s1.termination_reason == 2
    assert s0.termination_reason == s1.termination_reason and s0.num_iterations == s1.num_iterations
    n = s1.num_iterations
    assert np.array_equal(t0[:n, :], t1[:n, :])                     # identical until the last iteration
    assert t1[n, 6] == 1 and t0[n, 6] == 0                          # the last step: taken | not taken
    assert s0.num_successful_steps == s1.num_successful_steps - 1
    assert s0.final_cost == t1[n - 1, 0] and s1.final_cost == t1[n, 0]
    assert 0 <= s0.final_cost - s1.final_cost <= 1e-6 * s0.final_cost
    assert np.abs(o0.cams - o1.cams).max() < 1e-5


def test_non_finite_start_point_is_a_failure_before_any_step(O, scenes):
    """Ceres' rule (a residual block returning a non-finite value fails its evaluation; a failed INITIAL evaluation ends the solve
    as FAILURE, TrustRegionMinimizer::Init), in all three LM loops of the oracle; nothing is moved"""
    s = scenes.st20_scene(n_cams=8, n_pts=60, max_obs_per_pt=5, seed=3, pix_noise=1e-3)
    feat = s["obs_feat"].copy()
    feat[7, 0] = np.nan
    o = O.BA(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], feat, s["cam_fixed"])
    so, tr = o.solve(max_num_iterations=10)
    assert (so.termination_type, so.num_iterations, so.num_successful_steps) == (2, 0, 0) and np.isnan(so.initial_cost)
    assert np.array_equal(o.cams, s["cams0"]) and np.array_equal(o.pts, s["pts0"])
    x = np.linspace(0, 1, 20)

    def res(p):
        r = p[0] * x - 2.0 * x
        r[3] = np.inf
        return r, x[:, None]
    p, sd, _ = O.dense_lm(res, [0.5], 20)
    assert (sd.termination_type, sd.num_iterations) == (2, 0) and p[0] == 0.5
    g = scenes.pose_graph_scene(n_nodes=30, loops_per_node=2, seed=3)
    meas = g["meas"].copy()
    meas[5, 2] = np.nan
    pg = O.PG(g["poses0"], g["edge_i"], g["edge_j"], meas, g["node_fixed"])
    sp = pg.solve_sparse(max_num_iterations=5)[0]
    assert (sp.termination_type, sp.num_iterations) == (2, 0) and np.array_equal(pg.poses, g["poses0"])
