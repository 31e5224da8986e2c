"""Pose graph on the GPU (BASELINE config C4, build-defined).  Parity: HIP residual / Jacobian kernel
against the oracle element-wise; the LM + PCG solve against the oracle's LM + dense Cholesky on a graph
small enough for the dense oracle; the full 10k-node / 40k-edge configuration through its invariants
(cost decreases monotonically on accepted steps, node 0 fixed, ATE drops)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def st():
    mod = importlib.import_module("slam-tricks_amd")
    assert mod.device_count() > 0
    return mod


def test_pg_residual_jacobian_elementwise(st, O, scenes):
    s = scenes.pose_graph_scene(n_nodes=300, loops_per_node=3, seed=7, sigma_t=0.02, sigma_r=0.005, turns=6)
    e = st.PGEngine(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
    o = O.PG(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
    cost, r, Ji, Jj = e.evaluate()
    co, ro, Jio, Jjo = o.evaluate()
    assert abs(cost - co) <= 1e-12 * co
    assert np.abs(r - ro).max() < 1e-12
    assert np.abs(Ji - Jio).max() < 1e-11 and np.abs(Jj - Jjo).max() < 1e-11
    assert np.all(Ji[s["edge_i"] == 0] == 0)            # fixed node: columns dropped


def test_pg_solve_matches_dense_oracle(st, O, scenes):
    s = scenes.pose_graph_scene(n_nodes=150, loops_per_node=3, seed=4, sigma_t=0.02, sigma_r=0.004, turns=6)
    e = st.PGEngine(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
    o = O.PG(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
    summ, tr, pcg_total = e.solve()
    so, tro = o.solve()
    assert summ.termination_type == 0 and summ.num_iterations == so.num_iterations
    n = min(len(tr), len(tro))
    assert np.allclose(tr[:n, 0], tro[:n, 0], rtol=1e-7)
    assert np.all(tr[:n, 6] == tro[:n, 6])
    poses = e.get_poses()
    dq = np.minimum(np.abs(poses[:, :4] - o.poses[:, :4]).max(1), np.abs(poses[:, :4] + o.poses[:, :4]).max(1)).max()
    assert dq < 1e-7 and np.abs(poses[:, 4:] - o.poses[:, 4:]).max() < 1e-6
    assert O.pg_ate(s["poses_true"], poses) < 0.5 * O.pg_ate(s["poses_true"], s["poses0"])
    assert pcg_total > 0


def test_pg_config_c4_full_size(st, O, scenes):
    """10 000 SE3 nodes, ~40 000 edges"""
    s = scenes.pose_graph_scene(n_nodes=10000, loops_per_node=3, seed=4)
    assert 39000 <= len(s["edge_i"]) <= 40000
    e = st.PGEngine(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
    c0 = e.evaluate(jac=False)[0]
    summ, tr, pcg_total = e.solve(max_num_iterations=30)
    poses = e.get_poses()
    assert summ.final_cost < 0.05 * c0
    acc = tr[tr[:, 6] > 0, 0]
    assert np.all(np.diff(acc) <= 1e-12 * acc[:-1] + 1e-15)
    assert np.all(poses[0] == s["poses0"][0])
    ate0, ate1 = O.pg_ate(s["poses_true"], s["poses0"]), O.pg_ate(s["poses_true"], poses)
    assert ate1 < 0.2 * ate0
    # residual at the solution agrees with the oracle's evaluation of the same poses
    o = O.PG(poses, s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
    assert abs(o.evaluate(jac=False)[0] - summ.final_cost) <= 1e-9 * summ.final_cost
