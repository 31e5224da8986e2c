"""the two-view initialiser (st22-two-view/src/src/two_view_geometry.cpp:18-126).
CPU: the numpy oracle reproduces the st22 simulation.  GPU: the device path against the oracle."""
import importlib
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import two_view_np as TV  # noqa: E402

scenes = importlib.import_module("slam-tricks_amd.scenes")


def _same_up_to_sign(a, b, tol):
    s = np.sign(np.sum(a * b))
    return np.allclose(s * a, b, rtol=0, atol=tol * np.abs(b).max())


def test_oracle_reproduces_the_st22_simulation():
    """noise-free correspondences: pose = simulated pose up to the baseline length, points = simulated points
    (what st22 main.cpp:21-37 prints)"""
    s = scenes.two_view_pairs(n_pts=300, seed=22)
    r = TV.two_view_init(s["f1"], s["f2"], s["K"])
    assert r["R"] is not None and list(r["fails"] == 0).count(True) == 1
    assert np.allclose(r["R"], s["R_true"], atol=1e-8)
    scale = np.linalg.norm(s["t_true"])
    assert np.allclose(r["t"] * scale, s["t_true"], atol=1e-7)
    assert np.allclose(r["pts"] * scale, s["pts_f1"], rtol=0, atol=1e-6)
    # epipolar constraint with the convention x1^T F x2 = 0
    x1 = np.hstack([s["f1"], np.ones((300, 1))]); x2 = np.hstack([s["f2"], np.ones((300, 1))])
    assert np.abs(np.einsum("ni,ij,nj->n", x1, r["F"], x2)).max() < 1e-6 * np.abs(r["F"]).max() * 600 * 600


def test_oracle_survives_a_degenerate_configuration():
    """pure rotation (zero baseline): the epipolar system is rank-deficient and no translation direction is defined.
    What comes out is whatever the SVD gives (the reference behaves the same way); it must be well-formed: four
    hypothesis counters, and any pose that is returned is a proper rotation with a unit translation."""
    rng = np.random.default_rng(1)
    K = np.array([[400.0, 0, 300], [0, 400.0, 200], [0, 0, 1]])
    P = rng.uniform([-2, -2, 4], [2, 2, 9], (60, 3))
    a = 0.2
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    f1 = (P / P[:, 2:]) @ K.T
    P2 = P @ R                       # frame 2 = frame 1 rotated, same centre
    f2 = (P2 / P2[:, 2:]) @ K.T
    r = TV.two_view_init(f1[:, :2], f2[:, :2], K)
    assert len(r["fails"]) == 4 and all(0 <= int(k) <= len(P) for k in r["fails"])
    if r["R"] is not None:
        assert np.allclose(r["R"] @ r["R"].T, np.eye(3), atol=1e-8) and np.linalg.det(r["R"]) > 0.99
        assert abs(np.linalg.norm(r["t"]) - 1.0) < 1e-8
        assert np.count_nonzero(np.asarray(r["fails"]) == 0) == 1        # a pose is only returned for a unique winner


@pytest.mark.gpu
@pytest.mark.parametrize("n", [8, 300, 5000])
def test_device_path_matches_the_oracle(n):
    st = importlib.import_module("slam-tricks_amd")
    s = scenes.two_view_pairs(n_pts=max(n, 8), seed=22)
    f1, f2 = s["f1"][:n], s["f2"][:n]
    o = TV.two_view_init(f1, f2, s["K"])
    g = st.two_view_init(f1, f2, s["K"])
    assert _same_up_to_sign(g["F"], o["F"], 1e-7)
    # the ORDER of the four hypotheses depends on the sign conventions of the 3x3 SVD (LAPACK here, one-sided
    # Jacobi on the device, Eigen::JacobiSVD in the reference): the set is the same, and so is the winner
    assert np.array_equal(np.sort(g["fails"]), np.sort(o["fails"]))
    assert np.allclose(g["R"], o["R"], atol=1e-8) and np.allclose(g["t"], o["t"], atol=1e-8)
    assert np.allclose(g["pts"], o["pts"], rtol=1e-7, atol=1e-9)
    scale = np.linalg.norm(s["t_true"])
    assert np.allclose(g["R"], s["R_true"], atol=1e-7) and np.allclose(g["t"] * scale, s["t_true"], atol=1e-6)


@pytest.mark.gpu
def test_device_path_with_pixel_noise_matches_the_oracle():
    """noisy pixels (0.2 px): the unnormalised least-squares F of the reference, not a renormalised variant"""
    st = importlib.import_module("slam-tricks_amd")
    s = scenes.two_view_pairs(n_pts=2000, seed=5, pix_noise=0.2)
    o = TV.two_view_init(s["f1"], s["f2"], s["K"])
    try:
        g = st.two_view_init(s["f1"], s["f2"], s["K"])
    except st.StbaError:
        assert o["R"] is None
        return
    assert _same_up_to_sign(g["F"], o["F"], 1e-6)
    assert np.array_equal(np.sort(g["fails"]), np.sort(o["fails"]))
    if o["R"] is not None:
        assert np.allclose(g["R"], o["R"], atol=1e-7) and np.allclose(g["t"], o["t"], atol=1e-7)


@pytest.mark.gpu
def test_device_path_feeds_bundle_adjustment():
    """f1 -> the path: the two-view result seeds the C2 bundle adjustment, which converges to zero cost"""
    st = importlib.import_module("slam-tricks_amd")
    n = 2000
    s = scenes.two_view_pairs(n_pts=n, seed=22)
    g = st.two_view_init(s["f1"], s["f2"], s["K"])
    Kinv = np.linalg.inv(s["K"])
    x1 = (np.hstack([s["f1"], np.ones((n, 1))]) @ Kinv.T)[:, :2]
    x2 = (np.hstack([s["f2"], np.ones((n, 1))]) @ Kinv.T)[:, :2]
    cams = np.zeros((2, 7))
    cams[0, :4] = scenes.quat_from_rot(np.eye(3))
    cams[1, :4] = scenes.quat_from_rot(g["R"]); cams[1, 4:] = g["t"]
    obs_cam = np.tile(np.array([0, 1], dtype=np.int32), n)
    obs_pt = np.repeat(np.arange(n, dtype=np.int32), 2)
    feat = np.stack([x1, x2], 1).reshape(-1, 2)
    fixed = np.zeros((2, 6), dtype=np.uint8); fixed[0] = 1
    k = int(np.argmax(np.abs(g["t"])))
    fixed[1, 3 + k] = 1                               # gauge: one coordinate of the baseline
    e = st.BAEngine(cams, g["pts"], obs_cam, obs_pt, feat, fixed)
    summ, _ = e.solve()
    assert summ.initial_cost < 1e-12 and summ.final_cost < 1e-12
