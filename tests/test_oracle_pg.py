"""Pose graph (BASELINE config C4) -- build-defined: the reference has no pose-graph code, so the
oracle is pinned only by its own numerics: central differences of the residual, zero residual at
the truth, convergence, and the ATE metric of st4 (pose_simulation.cpp:198-209)."""
import numpy as np


def test_pg_jacobians_vs_central_differences(O, scenes):
    s = scenes.pose_graph_scene(n_nodes=40, loops_per_node=2, seed=1, sigma_t=0.02, sigma_r=0.01)
    pg = O.PG(s["poses0"], s["edge_i"], s["edge_j"], s["meas"])
    _, r, Ji, Jj = pg.evaluate()
    eps = 1e-6
    worst = 0.0
    for e in (0, 5, len(s["edge_i"]) - 1, len(s["edge_i"]) // 2):
        i, j = s["edge_i"][e], s["edge_j"][e]
        for which, node, Jan in ((0, i, Ji[e]), (1, j, Jj[e])):
            for k in range(6):
                d = np.zeros(6); d[k] = eps
                pp = s["poses0"].copy(); pp[node] = O.se3_retract(pp[node], d)
                pm = s["poses0"].copy(); pm[node] = O.se3_retract(pm[node], -d)
                rp = O.PG(pp, s["edge_i"], s["edge_j"], s["meas"]).evaluate(jac=False)[1][e]
                rm = O.PG(pm, s["edge_i"], s["edge_j"], s["meas"]).evaluate(jac=False)[1][e]
                num = (rp - rm) / (2 * eps)
                worst = max(worst, np.abs(num - Jan[:, k]).max())
    # truncated Jr^-1 series: error O(|r|^4); residuals here are ~0.1
    assert worst < 5e-5


def test_pg_zero_residual_at_truth_and_se3_algebra(O, scenes):
    s = scenes.pose_graph_scene(n_nodes=30, loops_per_node=2, seed=2, sigma_t=0.0, sigma_r=0.0)
    pg = O.PG(s["poses_true"], s["edge_i"], s["edge_j"], s["meas"])
    cost, r, _, _ = pg.evaluate(jac=False)
    assert cost < 1e-24
    a, b = s["poses_true"][3], s["poses_true"][17]
    ident = O.se3_compose(a, O.se3_inverse(a))
    assert np.allclose(ident, [0, 0, 0, 1, 0, 0, 0], atol=1e-14)
    assert np.allclose(O.se3_compose(O.se3_compose(a, b), O.se3_inverse(b)), a, atol=1e-13)
    assert O.pg_ate(s["poses_true"], s["poses_true"]) < 1e-14


def test_pg_solve_reduces_ate(O, scenes):
    s = scenes.pose_graph_scene(n_nodes=120, loops_per_node=3, seed=4, sigma_t=0.02, sigma_r=0.004, turns=6)
    pg = O.PG(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
    ate0 = O.pg_ate(s["poses_true"], s["poses0"])
    summ, tr = pg.solve()
    ate1 = O.pg_ate(s["poses_true"], pg.poses)
    assert summ.termination_type == 0 and summ.final_cost < summ.initial_cost * 0.2
    assert ate1 < 0.5 * ate0
    assert np.all(pg.poses[0] == s["poses0"][0])        # node 0 fixed


def test_pg_matrix_free_lm_reproduces_the_dense_lm_trace(O, scenes):
    """orc_pg_solve_sparse (certified conjugate gradients on the matrix-free normal equations) against orc_pg_solve (dense
    normal equations + Cholesky): the same LM, iteration for iteration."""
    s = scenes.pose_graph_scene(n_nodes=150, loops_per_node=3, seed=4, sigma_t=0.02, sigma_r=0.004, turns=6)
    a = O.PG(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
    b = O.PG(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
    sa, tra = a.solve()
    sb, trb, cg, worst = b.solve_sparse()
    assert sa.termination_type == 0 and sb.termination_type == 0
    assert sa.num_iterations == sb.num_iterations and sa.termination_reason == sb.termination_reason
    assert np.allclose(tra[:, 0], trb[:, 0], rtol=1e-11) and np.all(tra[:, 6] == trb[:, 6])
    assert np.allclose(tra[:, 5], trb[:, 5], rtol=1e-9)            # radius
    assert np.abs(a.poses - b.poses).max() < 1e-11
    assert cg > 0 and worst <= 1e-10


def test_pg_matrix_free_first_step_against_a_sparse_direct_solve(O, scenes):
    """one LM step of a 1 500-node graph: the oracle's certified CG step against scipy's sparse LU of the same damped
    normal equations (an independent solver: the new cost must agree)"""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    s = scenes.pose_graph_scene(n_nodes=1500, loops_per_node=3, seed=9, sigma_t=0.02, sigma_r=0.004, turns=6)
    ei, ej = s["edge_i"], s["edge_j"]
    n, m = 1500, len(ei)
    pg = O.PG(s["poses0"], ei, ej, s["meas"], s["node_fixed"])
    cost, r, Ji, Jj = pg.evaluate()
    rows = np.repeat(np.arange(6 * m).reshape(m, 6, 1), 6, 2)
    ci = 6 * ei[:, None, None] + np.arange(6)[None, None, :] + np.zeros((m, 6, 1), int)
    cj = 6 * ej[:, None, None] + np.arange(6)[None, None, :] + np.zeros((m, 6, 1), int)
    J = sp.csr_matrix((np.concatenate([Ji.ravel(), Jj.ravel()]), (np.concatenate([rows.ravel()] * 2), np.concatenate([ci.ravel(), cj.ravel()]))),
                      shape=(6 * m, 6 * n))
    H = (J.T @ J).tocsc()
    g = J.T @ r.ravel()
    diag = H.diagonal()
    scale = 1.0 / (1.0 + np.sqrt(diag))
    D = np.clip(diag * scale ** 2, 1e-6, 1e32) / 1e4 / scale ** 2
    dx = spl.splu((H + sp.diags(D)).tocsc()).solve(-g)
    newp = pg.poses.copy()
    for k in range(n):
        if not s["node_fixed"][k]:
            newp[k] = O.se3_retract(pg.poses[k], dx[6 * k:6 * k + 6])
    c_direct = O.PG(newp, ei, ej, s["meas"], s["node_fixed"]).evaluate(jac=False)[0]
    opt = O.default_options(max_num_iterations=1)
    summ, tr, cg, worst = pg.solve_sparse(opt)
    assert worst <= 1e-10
    assert abs(tr[1, 0] - c_direct) <= 1e-9 * c_direct
    assert np.abs(pg.poses - newp).max() < 1e-9


def test_c4_oracle_has_not_drifted_from_its_frozen_trace(O, scenes):
    """tests/golden/oracle_traces.json["c4"]: the C4-size oracle (10 000 nodes, matrix-free LM) against what it produced when
    the fixture was made -- and the fixture's generator held its first step against a sparse direct solve of the same system"""
    import json
    import os
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "oracle_traces.json")) as f:
        g = json.load(f)["c4"]
    s = scenes.pose_graph_scene(n_nodes=10000, loops_per_node=3, seed=4)
    o = O.PG(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
    assert (o.n, o.ne) == (g["n_nodes"], g["n_edges"])
    summ, tr, cg, worst = o.solve_sparse()
    assert summ.num_iterations == g["num_iterations"] and summ.termination_reason == g["termination_reason"]
    assert [int(x) for x in tr[:, 6]] == g["accepted"]
    assert np.allclose(tr[:, 0], g["cost_trace"], rtol=1e-9)
    assert worst <= 1e-10
    assert abs(g["cost_trace"][1] - g["first_trial_cost_sparse_direct"]) <= 1e-9 * g["cost_trace"][1]
    assert np.abs(o.poses[::50].reshape(-1) - np.array(g["final_poses_every_50th"])).max() < 1e-7
    assert abs(O.pg_ate(s["poses_true"], o.poses) - g["ate_final"]) < 1e-9
