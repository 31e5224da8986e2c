"""odometry files (st16-pcl-viewer/src/src/scene.cpp:66-110) and the absolute trajectory error
(st4-kalman/src/src/pose_simulation.cpp:198-209) behind the C ABI; the reference's own 70-pose lidar
odometry file is the fixture (tests/golden/st16_odom, copied by make_golden.py)."""
import importlib
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
GOLDEN = os.path.join(HERE, "golden", "st16_odom", "odometryInfo.txt")


@pytest.fixture(scope="module")
def st():
    return importlib.import_module("slam-tricks_amd")


def _parse(path):
    """the reference's reader restated: stod for stamp and quaternion, normalise, stof for the translation"""
    lines = open(path).read().splitlines()
    n = int(lines[1].split(" ")[2])
    k = next(i for i, ln in enumerate(lines) if ln.startswith("end_header")) + 1
    stamps, poses = np.zeros(n), np.zeros((n, 7))
    for i in range(n):
        v = lines[k + i].split(" ")
        stamps[i] = float(v[0])
        q = np.array([float(x) for x in v[1:5]])
        poses[i, :4] = q / np.linalg.norm(q)
        poses[i, 4:] = [float(np.float32(x)) for x in v[5:8]]
    return stamps, poses


def test_odometry_file_parses_like_the_reference(st):
    stamps, poses = st.odometry_read(GOLDEN)
    rs, rp = _parse(GOLDEN)
    assert poses.shape == (70, 7)
    assert np.array_equal(stamps, rs)
    assert np.allclose(poses[:, :4], rp[:, :4], rtol=0, atol=1e-15) and np.array_equal(poses[:, 4:], rp[:, 4:])
    assert np.allclose(np.linalg.norm(poses[:, :4], axis=1), 1.0, atol=1e-15)


def test_odometry_round_trip_and_errors(st, tmp_path):
    stamps, poses = st.odometry_read(GOLDEN)
    p = str(tmp_path / "odom.txt")
    st.odometry_write(p, stamps, poses)
    s2, p2 = st.odometry_read(p)
    assert np.allclose(s2, stamps, rtol=0, atol=1e-6)            # 9 decimals of a 1.6e9 s stamp
    assert np.allclose(p2[:, :4], poses[:, :4], atol=2e-10) and np.allclose(p2[:, 4:], poses[:, 4:], atol=1e-6)
    with pytest.raises(st.StbaError):
        st.odometry_read(str(tmp_path / "missing.txt"))
    bad = tmp_path / "short.txt"
    bad.write_text("format ascii 1.0\nelement odometryInfo 3\nend_header\n1 0 0 0 1 0 0 0\n")
    with pytest.raises(st.StbaError):
        st.odometry_read(str(bad))


def test_ate_matches_the_oracle(st):
    import oracle_py as O
    scenes = importlib.import_module("slam-tricks_amd.scenes")
    s = scenes.pose_graph_scene(n_nodes=200, loops_per_node=2, seed=4)
    a, b = s["poses_true"], s["poses0"]
    assert abs(st.trajectory_ate(a, b) - O.pg_ate(a, b)) <= 1e-12 * max(1.0, O.pg_ate(a, b))
    assert st.trajectory_ate(a, a) < 1e-14
    flipped = b.copy(); flipped[:, :4] *= -1                      # q and -q are the same rotation
    assert abs(st.trajectory_ate(a, flipped) - st.trajectory_ate(a, b)) < 1e-12


@pytest.mark.gpu
def test_odometry_file_feeds_the_pose_graph(st):
    """file -> pose graph: the 70 lidar poses as truth, exact relative measurements between neighbours and
    every fifth pair, a drifting start; the device solver brings the ATE back to ~0"""
    scenes = importlib.import_module("slam-tricks_amd.scenes")
    _, truth = st.odometry_read(GOLDEN)
    n = len(truth)
    ei = np.concatenate([np.arange(n - 1), np.arange(0, n - 5, 5)]).astype(np.int32)
    ej = np.concatenate([np.arange(1, n), np.arange(5, n, 5)]).astype(np.int32)
    meas = scenes._se3_mul(scenes._se3_inv(truth[ei]), truth[ej])
    rng = np.random.default_rng(16)
    init = truth.copy()
    for i in range(1, n):                                         # integrate noisy odometry
        step = scenes._se3_mul(scenes._se3_inv(truth[i - 1:i]), truth[i:i + 1])[0]
        noisy = scenes._se3_mul(step[None], scenes._se3_exp7(rng.normal(0, [0.02] * 3 + [0.01] * 3))[None])[0]
        init[i] = scenes._se3_mul(init[i - 1:i], noisy[None])[0]
    fixed = np.zeros(n, np.uint8); fixed[0] = 1
    e = st.PGEngine(init, ei, ej, meas, fixed)
    ate0 = st.trajectory_ate(truth, init)
    summ, _, _ = e.solve()
    ate1 = st.trajectory_ate(truth, e.get_poses())
    assert ate0 > 0.05 and ate1 < 1e-6 and summ.final_cost < 1e-12
