"""CPU tests of the oracle's primitives (no GPU): SO3/SE3, the reprojection factor and its
Jacobian against central differences and against the autodiff-path composition
d r/d q (ambient) x Dx_this_mul_exp_x_at_0 (solver.hpp:48-54, notes.tex:131-144)."""
import numpy as np
import pytest


def num_jac(f, x, eps=1e-6):
    x = np.asarray(x, float)
    f0 = f(x)
    J = np.zeros((f0.size, x.size))
    for i in range(x.size):
        d = np.zeros_like(x); d[i] = eps
        J[:, i] = (f(x + d) - f(x - d)) / (2 * eps)
    return J


def rand_pose(rng, O):
    q = O.so3_exp(rng.normal(0, 1.0, 3))
    t = rng.normal(0, 1.0, 3)
    return q, t


def test_so3_exp_log_roundtrip(O):
    rng = np.random.default_rng(0)
    for _ in range(50):
        w = rng.normal(0, 1.0, 3)
        q = O.so3_exp(w)
        assert abs(np.linalg.norm(q) - 1) < 1e-14
        assert np.allclose(O.so3_log(q), w, atol=1e-12)
    assert np.allclose(O.so3_exp(np.zeros(3)), [0, 0, 0, 1])
    w = np.array([1e-12, -2e-12, 3e-12])
    assert np.allclose(O.so3_log(O.so3_exp(w)), w, atol=1e-24, rtol=1e-9)


def test_quat_rot_roundtrip(O, scenes):
    rng = np.random.default_rng(1)
    for _ in range(20):
        q, _ = rand_pose(rng, O)
        R = O.quat_to_rot(q)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-14)
        assert np.allclose(R, scenes.rot_from_quat(q), atol=1e-15)
        q2 = O.rot_to_quat(R)
        assert min(np.abs(q2 - q).max(), np.abs(q2 + q).max()) < 1e-14


def test_plus_jacobian_matches_notes_and_numeric(O):
    rng = np.random.default_rng(2)
    q, _ = rand_pose(rng, O)
    J = O.so3_plus_jacobian(q)
    x, y, z, w = q
    expect = 0.5 * np.array([[w, -z, y], [z, w, -x], [-y, x, w], [-x, -y, -z]])   # notes.tex:131-144
    assert np.allclose(J, expect, atol=1e-16)
    Jn = num_jac(lambda d: O.so3_plus(q, d), np.zeros(3), 1e-6)
    assert np.allclose(J, Jn, atol=1e-9)


def test_r3_plus(O):
    rng = np.random.default_rng(3)
    x = rng.normal(0, 0.7, 3); d = rng.normal(0, 0.1, 3)
    got = O.so3r3_plus(x, d)                                    # solver.hpp:67-78
    R = O.quat_to_rot(O.so3_exp(x)) @ O.quat_to_rot(O.so3_exp(d))
    assert np.allclose(O.quat_to_rot(O.so3_exp(got)), R, atol=1e-13)


def test_se3_exp_log(O, scenes):
    rng = np.random.default_rng(4)
    for _ in range(20):
        xi = rng.normal(0, 0.8, 6)
        q, t = O.se3_exp(xi)
        R, t2 = scenes.se3_exp(xi)
        assert np.allclose(O.quat_to_rot(q), R, atol=1e-13) and np.allclose(t, t2, atol=1e-13)
        assert np.allclose(O.se3_log(q, t), xi, atol=1e-11)
        assert np.allclose(scenes.se3_log(R, t2), xi, atol=1e-10)


def test_reproj_jacobian_vs_central_differences(O):
    """hat(pInC) is right; the reference's solver.hpp:195 formula is not."""
    rng = np.random.default_rng(5)
    worst_ok, worst_ref = 0.0, 0.0
    for _ in range(30):
        q, t = rand_pose(rng, O)
        L = t + O.quat_to_rot(q) @ (np.array([0, 0, 4.0]) + rng.normal(0, 1.0, 3))
        f = rng.normal(0, 0.1, 2)

        def fr(d):
            return O.reproj_residual(O.so3_plus(q, d[:3]), t + d[3:6], L + d[6:9], f)
        Jn = num_jac(fr, np.zeros(9), 1e-6)
        Jc, Jp = O.reproj_jacobian(q, t, L, 0)
        worst_ok = max(worst_ok, np.abs(np.hstack([Jc, Jp]) - Jn).max())
        Jc1, _ = O.reproj_jacobian(q, t, L, 1)
        worst_ref = max(worst_ref, np.abs(Jc1[:, :3] - Jn[:, :3]).max())
        assert np.allclose(Jc1[:, 3:], Jc[:, 3:])               # translation block is right in both
    assert worst_ok < 1e-7
    assert worst_ref > 1e-2                                     # the reference formula is wrong off t = 0


def test_autodiff_composition_equals_analytic(O):
    """ambient d r / d q (2x4, what Jets give for test_ceres.h:63-80) times the 4x3 plus Jacobian
    equals the analytic 2x3 rotation block."""
    rng = np.random.default_rng(6)
    for _ in range(20):
        q, t = rand_pose(rng, O)
        L = t + O.quat_to_rot(q) @ (np.array([0.3, -0.2, 5.0]) + rng.normal(0, 0.5, 3))
        f = np.zeros(2)
        assert np.allclose(O.reproj_residual(q, t, L, f), O.reproj_residual(q, t, L, f, ambient=True), atol=1e-14)
        Jq = num_jac(lambda qq: O.reproj_residual(qq, t, L, f, ambient=True), q, 1e-6)
        Jloc = Jq @ O.so3_plus_jacobian(q)
        Jc, _ = O.reproj_jacobian(q, t, L, 0)
        assert np.allclose(Jloc, Jc[:, :3], atol=1e-7)


def test_cholesky(O):
    rng = np.random.default_rng(7)
    for n in (1, 5, 64, 65, 200, 333):
        A = rng.normal(size=(n, n)); A = A @ A.T + n * np.eye(n)
        rc, L = O.cholesky_lower(A, threads=2)
        assert rc == 0
        Lr = np.linalg.cholesky(A)
        assert np.allclose(np.tril(L), Lr, atol=1e-10)
        b = rng.normal(size=n)
        assert np.allclose(O.cholesky_solve(L, b), np.linalg.solve(A, b), atol=1e-9)
    rc, _ = O.cholesky_lower(np.array([[1.0, 2.0], [2.0, 1.0]]))
    assert rc != 0
