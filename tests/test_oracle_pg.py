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
