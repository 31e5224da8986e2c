"""Pins the oracle against every known answer the reference holds for this path
(SURVEY.md section 4 / tests/golden/known_answers.json).  CPU only."""
import os

import numpy as np

from conftest import GOLDEN
import zhang_init as Z


def _csv(name):
    return np.loadtxt(os.path.join(GOLDEN, name), delimiter=",")


def test_parabola_known_answers(O, known):
    ka = known["st7_parabola"]
    good, bad = _csv("st7_ransac/good.csv"), _csv("st7_ransac/bad.csv")
    assert good.shape == (100, 2) and bad.shape == (200, 2)
    # values are printed with 6 significant digits (drawerResult.py:12-16)
    assert np.allclose(O.parabola_least_square(good), ka["leastSquare_good"], rtol=2e-5)
    assert np.allclose(O.parabola_least_square(bad), ka["leastSquare_bad"], rtol=2e-5)
    g, _ = O.parabola_gauss_newton(good, 10)
    b, _ = O.parabola_gauss_newton(bad, 10)
    assert np.allclose(g, ka["gaussNewton_good_10"], rtol=2e-5)
    assert np.allclose(b, ka["gaussNewton_bad_10"], rtol=2e-5)


def test_se2_gauss_newton_iterates_of_the_icp_demo(O):
    """st6-icp/log/binding: the reference's recorded inputs (pc1, pc2) and its iterates after ONE and TWO Gauss-Newton
    steps on SE2 with the left-multiplicative update T <- exp(delta) T (icp.hpp:28-50; binding.cpp applies the estimate
    to pc1 and writes pc1_prime_k.csv).  The only reference-held per-iteration trace of a manifold Gauss-Newton: the
    oracle's float restatement reproduces both iterates to the 6 digits the files carry."""
    pc1, pc2 = _csv("st6_icp/pc1.csv"), _csv("st6_icp/pc2.csv")
    assert pc1.shape == (10, 2) and pc2.shape == (10, 2)
    for k in (1, 2):
        want = _csv(f"st6_icp/pc1_prime_{k}.csv")
        c, s, tx, ty = (float(v) for v in O.icp_se2_gauss_newton(pc1, pc2, k))
        got = pc1 @ np.array([[c, s], [-s, c]]) + np.array([tx, ty])
        assert np.abs(got - want).max() < 5e-5 * max(1.0, np.abs(want).max()), (k, np.abs(got - want).max())
    # and the loop converges to the transformation the demo was generated with (pi/4, (2, 2)) up to its 0.1 noise
    c, s, tx, ty = (float(v) for v in O.icp_se2_gauss_newton(pc1, pc2, 10))
    assert abs(np.arctan2(s, c) - np.pi / 4) < 0.02 and abs(tx - 2.0) < 0.15 and abs(ty - 2.0) < 0.15


def test_parabola_through_dense_lm(O, known):
    """the same fit through the LM driver (FP64) lands on the same answer"""
    good = _csv("st7_ransac/good.csv")
    x, y = good[:, 0], good[:, 1]

    def res(p):
        return p[0] * x * x + p[1] * x + p[2] - y, np.stack([x * x, x, np.ones_like(x)], 1)
    p, summ, _ = O.dense_lm(res, [1.0, 0.0, 0.0], len(x))
    assert summ.termination_type == 0
    assert np.allclose(p, known["st7_parabola"]["gaussNewton_good_10"], rtol=5e-5)


def test_ceres_bound_demo(O, known):
    ka = known["st17_ceres_bound"]

    def res(p):
        return np.array([p[0] - 3.0]), np.array([[1.0]])          # ceres_bound.cpp:19
    x, summ, _ = O.dense_lm(res, [ka["x0"]], 1)
    assert abs(x[0] - ka["x_free"]) < 1e-8 and summ.termination_type == 0
    x, summ, _ = O.dense_lm(res, [ka["x0"]], 1, lower=[ka["lower"]], upper=[ka["upper"]])
    assert abs(x[0] - ka["x_bounded"]) < 1e-12 and summ.termination_type == 0


def test_published_pnp_poses(scenes, known):
    ka = known["st17_pnp"]
    real, init = scenes.pnp_published_poses()
    for pose, q, t in ((real, ka["q_true_xyzw"], ka["t_true"]), (init, ka["q_init_xyzw"], ka["t_init"])):
        qq = np.array(q)
        assert min(np.abs(pose[:4] - qq).max(), np.abs(pose[:4] + qq).max()) < 6e-6   # 5 printed digits
        assert np.allclose(pose[4:], t)


def test_pnp_gauss_newton_converges_to_published_truth(O, scenes, known):
    """SelfGaussNewton (solver.hpp:387-462) from the published init reaches the published truth;
    the reference's rotation Jacobian (rot_mode=1) reaches the same fixed point in more
    iterations (release.png: SizedCostFunction 8 it. vs autodiff 6)."""
    s = scenes.pnp_scene(seed=17)
    assert 10 <= len(s["pts"]) <= 40
    q, t, it0, _ = O.pnp_gauss_newton(s["pts"], s["feats"], s["pose_init"][:4], s["pose_init"][4:], 0)
    qt = s["pose_true"][:4]
    assert min(np.abs(q - qt).max(), np.abs(q + qt).max()) < 1e-9
    assert np.allclose(t, s["pose_true"][4:], atol=1e-8)
    q1, t1, it1, _ = O.pnp_gauss_newton(s["pts"], s["feats"], s["pose_init"][:4], s["pose_init"][4:], 1, max_iter=40)
    assert min(np.abs(q1 - qt).max(), np.abs(q1 + qt).max()) < 1e-7
    assert it1 > it0
    assert it0 <= known["st17_pnp"]["release_iterations"]["SelfGaussNewton"] + 1


def test_pnp_lm_as_ba_with_fixed_landmarks(O, scenes, known):
    """SolvePnPWith*: one camera, landmarks constant, Ceres-style LM; cost -> ~0 in a handful of
    iterations as in release.png (6 it., final cost 4e-21)."""
    s = scenes.pnp_scene(seed=17)
    n = len(s["pts"])
    ba = O.BA(s["pose_init"][None], s["pts"], np.zeros(n, np.int32), np.arange(n, dtype=np.int32), s["feats"],
              pt_fixed=np.ones(n, np.uint8))
    summ, tr = ba.solve()
    assert summ.termination_type == 0
    assert summ.final_cost < known["st17_pnp"]["final_cost_below"]
    assert 4 <= summ.num_iterations <= 9
    qt = s["pose_true"][:4]
    assert min(np.abs(ba.cams[0, :4] - qt).max(), np.abs(ba.cams[0, :4] + qt).max()) < 1e-9


def test_calibration_fixture(O, known):
    """st3-calibration/calib/1..9.txt: Zhang init + totalOptimization reproduce the recorded
    intrinsics and cost trace."""
    ka = known["st3_calibration"]
    obj, img = Z.read_corners(os.path.join(GOLDEN, "st3_calib"), ka["board_square_m"])
    assert obj.shape == (9, 40, 2)
    p0 = Z.zhang_init(obj, img, lambda R, t: O.se3_log(O.rot_to_quat(R), t))
    assert np.allclose(p0[:4], ka["init_fx_fy_u0_v0"], rtol=2e-7)
    p, it, sse = O.calib_gauss_newton(p0, obj, img, 10)
    assert abs(sse[0] - ka["sse_first"]) < 5e-4
    done = sse[~np.isnan(sse)]
    assert abs(done[-1] - ka["sse_last"]) < 5e-4
    assert np.allclose(p[:4], ka["final_intr_dist"][:4], rtol=0, atol=6e-4)      # printed to 3 decimals
    assert np.allclose(p[4:7], ka["final_intr_dist"][4:7], rtol=2e-6)
    assert np.allclose(p[7:9], ka["final_intr_dist"][7:9], rtol=2e-3, atol=2e-9)
    assert abs(it - ka["gn_iterations"]) <= 1


def test_calibration_jacobian_numeric(O, scenes):
    s = scenes.calib_scene(n_views=3, rows=3, cols=4, seed=5)
    params = np.concatenate([s["intr_true"], s["xis_true"].reshape(-1)])
    _, e, Ji, Jx = O.calib_evaluate(params, s["obj"], s["img"])
    eps = 1e-6
    for k in range(9):
        d = np.zeros_like(params); d[k] = eps * max(1.0, abs(params[k]))
        ep = O.calib_evaluate(params + d, s["obj"], s["img"], jac=False)[1]
        em = O.calib_evaluate(params - d, s["obj"], s["img"], jac=False)[1]
        num = (ep - em) / (2 * d[k])
        assert np.allclose(num, Ji[..., k], rtol=2e-5, atol=2e-4)
    # left perturbation exp(d) * T of view 1 (calib.cpp:397-402)
    v = 1
    for k in range(6):
        d = np.zeros(6); d[k] = eps
        def shifted(sign):
            Rd, td = scenes.se3_exp(sign * d)
            R, t = scenes.se3_exp(params[9 + 6 * v: 15 + 6 * v])
            pp = params.copy()
            pp[9 + 6 * v: 15 + 6 * v] = scenes.se3_log(Rd @ R, Rd @ t + td)
            return O.calib_evaluate(pp, s["obj"], s["img"], jac=False)[1]
        num = (shifted(+1) - shifted(-1)) / (2 * eps)
        assert np.allclose(num[v], Jx[v, ..., k], rtol=1e-5, atol=1e-3)


def test_published_iteration_counts_against_the_oracles_distribution(O, scenes):
    """The only iteration counts the reference publishes: st17-ceres/img/release.png (20 valid features, initial cost 2.232755:
    `Iterations: 6` for both autodiff solvers, `iter num: 7` for SelfGaussNewton) and img/debug.png (21 features, initial cost
    2.958739: `Iterations: 7`, `iter num: 7`).  The scenes behind the screenshots are clock-seeded and cannot be reproduced, but
    scenes LIKE them can be drawn from the same generator (main.cpp:37-87 restated in scenes.pnp_scene): same feature count,
    initial cost within 25 % of the published one.  Over those scenes:
      * SelfGaussNewton with the reference's rotation Jacobian (solver.hpp:387-462, :425) takes 7 iterations most often -- the
        published count in both screenshots;
      * the oracle's Ceres-style LM takes 5 or 6 steps (release-like) and 6 steps (debug-like, 9 of 10 scenes).  Ceres'
        BriefReport counts the initial evaluation as an iteration (Ceres >= 2.0: "Iterations" = successful + unsuccessful steps with
        iteration 0 among them), so the published 6 / 7 are 5 / 6 LM steps: both inside what the oracle does, the second one its
        mode.  Read as plain step counts, the debug screenshot's 7 would be a count the oracle NEVER takes on such scenes -- the
        two screenshots together decide the reading.
    A statistical pin of the oracle's iteration behaviour against the reference's published runs (a single scene could agree
    by luck).  It is not a per-iteration diff against Ceres, which stays unpinned."""
    import collections
    dist = {}
    for tag, n_feat, c0 in (("release", 20, 2.232755), ("debug", 21, 2.958739)):
        lm, gn = collections.Counter(), collections.Counter()
        for seed in range(900):
            s = scenes.pnp_scene(seed=seed)
            n = len(s["pts"])
            if n != n_feat:
                continue
            ba = O.BA(s["pose_init"][None], s["pts"], np.zeros(n, np.int32), np.arange(n, dtype=np.int32), s["feats"],
                      pt_fixed=np.ones(n, np.uint8))
            summ, _ = ba.solve()
            if abs(summ.initial_cost - c0) > 0.25 * c0:
                continue
            assert summ.termination_type == 0 and summ.final_cost < 1e-15
            lm[summ.num_iterations] += 1
            _, _, it_ref, _ = O.pnp_gauss_newton(s["pts"], s["feats"], s["pose_init"][:4], s["pose_init"][4:], 1, max_iter=40)
            gn[it_ref] += 1
        dist[tag] = (lm, gn, sum(lm.values()))
        assert dist[tag][2] >= 25, dist
    lm_r, gn_r, tot_r = dist["release"]
    lm_d, gn_d, tot_d = dist["debug"]
    # SelfGaussNewton, published 7 and 7
    assert gn_r.most_common(1)[0][0] == 7 and gn_r[7] >= 0.6 * tot_r, gn_r
    assert gn_d[7] >= 0.4 * tot_d and set(gn_d) <= {6, 7, 8, 9}, gn_d
    # LM: published "Iterations" 6 and 7 = 5 and 6 steps
    assert set(lm_r) <= {5, 6, 7} and lm_r[5] >= 0.2 * tot_r, lm_r
    assert lm_d.most_common(1)[0][0] == 6 and lm_d[6] >= 0.7 * tot_d, lm_d
    assert lm_d[7] == 0          # (the plain-step reading of the debug screenshot is outside the oracle's behaviour)
