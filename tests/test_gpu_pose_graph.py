"""Pose graph on the GPU (BASELINE config C4, build-defined).  Parity: HIP residual / Jacobian kernel against the oracle
element-wise; the LM solve with EXACT steps (PCG to 1e-12) against the oracle's LM trace for trace -- on a graph small enough
for the dense oracle and, since round 4, at the FULL C4 size against the frozen trace of the oracle's matrix-free LM
(tests/golden/oracle_traces.json["c4"]); the production solve (inexact Newton steps, forcing sequence) against the same
oracle on converged quantities: final cost 1e-6, poses 1e-5, ATE."""
import importlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def st():
    mod = importlib.import_module("slam-tricks_amd")
    assert mod.device_count() > 0
    return mod


@pytest.fixture(scope="module")
def c4(scenes):
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "oracle_traces.json")) as f:
        gold = json.load(f)["c4"]
    s = scenes.pose_graph_scene(n_nodes=10000, loops_per_node=3, seed=4)
    assert (len(s["poses0"]), len(s["edge_i"])) == (gold["n_nodes"], gold["n_edges"])
    return s, gold


def pose_diff(a, b):
    dq = np.minimum(np.abs(a[:, :4] - b[:, :4]).max(1), np.abs(a[:, :4] + b[:, :4]).max(1)).max()
    return max(dq, np.abs(a[:, 4:] - b[:, 4:]).max())


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


@pytest.mark.parametrize("group", [0, 16, -1])
def test_pg_exact_steps_follow_the_dense_oracle(st, O, scenes, group):
    """fixed PCG tolerance 1e-12 (exact LM steps): iteration for iteration the oracle's dense LM -- with the automatic coarse
    space, a coarser one, and without (block Jacobi only: the preconditioner must not change the answer)"""
    s = scenes.pose_graph_scene(n_nodes=150, loops_per_node=3, seed=4, sigma_t=0.02, sigma_r=0.004, turns=6)
    e = st.PGEngine(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
    o = O.PG(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
    summ, tr, pcg_total = e.solve(pcg=e.pcg_options(forcing_eta0=0.0, coarse_group=group))
    so, tro = o.solve()
    assert summ.termination_type == 0 and summ.num_iterations == so.num_iterations
    n = min(len(tr), len(tro))
    assert np.allclose(tr[:n, 0], tro[:n, 0], rtol=1e-7)
    assert np.all(tr[:n, 6] == tro[:n, 6])
    poses = e.get_poses()
    assert pose_diff(poses, o.poses) < 1e-6
    assert O.pg_ate(s["poses_true"], poses) < 0.5 * O.pg_ate(s["poses_true"], s["poses0"])
    ps = e.pcg_summary()
    assert ps.iterations_total == pcg_total > 0 and ps.hit_cap == 0 and ps.solves == summ.num_iterations
    assert (ps.coarse_dim > 0) == (group >= 0)


def test_pg_coarse_space_cuts_the_pcg_iterations(st, scenes):
    """the point of the two-level preconditioner: on a 3 000-node spiral the same exact-step solve needs several times fewer
    PCG iterations with the rigid-body coarse space than with block Jacobi alone"""
    s = scenes.pose_graph_scene(n_nodes=3000, loops_per_node=3, seed=5, turns=8)
    its = {}
    for group in (0, -1):
        e = st.PGEngine(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
        summ, tr, total = e.solve(max_num_iterations=3, pcg=e.pcg_options(forcing_eta0=0.0, relative_tolerance=1e-10, coarse_group=group,
                                                                          max_iterations=4000))
        its[group] = total
        assert e.pcg_summary().hit_cap == 0
    assert its[0] * 4 < its[-1], its


def test_pg_inexact_steps_reach_the_oracles_answer(st, O, scenes):
    """production options (forcing sequence eta0 = 0.1): same converged cost (1e-6) and poses (1e-5) as the exact-step oracle"""
    s = scenes.pose_graph_scene(n_nodes=300, loops_per_node=3, seed=11, sigma_t=0.02, sigma_r=0.004, turns=6)
    e = st.PGEngine(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
    o = O.PG(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
    tight = dict(function_tolerance=1e-12, parameter_tolerance=1e-11)
    summ, tr, pcg_total = e.solve(**tight)
    so, tro, _, _ = o.solve_sparse(**tight)
    assert summ.termination_type == 0 and so.termination_type == 0
    assert abs(summ.final_cost - so.final_cost) <= 1e-6 * so.final_cost
    assert pose_diff(e.get_poses(), o.poses) < 1e-5
    assert e.pcg_summary().hit_cap == 0


def test_pg_config_c4_exact_steps_follow_the_frozen_oracle_trace(st, O, c4):
    """10 000 SE3 nodes, 39 999 edges, exact LM steps: the oracle's trace at FULL size (frozen, make_oracle_traces.py)"""
    s, gold = c4
    e = st.PGEngine(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
    summ, tr, pcg_total = e.solve(pcg=e.pcg_options(forcing_eta0=0.0, relative_tolerance=1e-12, max_iterations=2000))
    assert summ.num_iterations == gold["num_iterations"] and summ.termination_reason == gold["termination_reason"]
    assert [int(x) for x in tr[:, 6]] == gold["accepted"]
    assert np.allclose(tr[:, 0], gold["cost_trace"], rtol=1e-7)
    assert np.allclose(tr[:, 5], gold["radius_trace"], rtol=1e-6)
    poses = e.get_poses()
    assert pose_diff(poses[::50], np.array(gold["final_poses_every_50th"]).reshape(-1, 7)) < 1e-5
    ps = e.pcg_summary()
    assert ps.hit_cap == 0 and ps.iterations_total < 250 * summ.num_iterations      # (round 3: ~993 per LM iteration = the cap)


def test_pg_config_c4_full_size(st, O, c4):
    """the production solve at C4: inexact steps, <= 100 PCG iterations per LM iteration, and the oracle's converged answer"""
    s, gold = c4
    e = st.PGEngine(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
    c0 = e.evaluate(jac=False)[0]
    summ, tr, pcg_total = e.solve()
    poses = e.get_poses()
    assert summ.termination_type == 0 and abs(summ.num_iterations - gold["num_iterations"]) <= 1
    assert abs(c0 - gold["initial_cost"]) <= 1e-12 * c0
    assert abs(summ.final_cost - gold["final_cost"]) <= 1e-6 * gold["final_cost"]
    assert pose_diff(poses[::50], np.array(gold["final_poses_every_50th"]).reshape(-1, 7)) < 1e-5
    acc = tr[tr[:, 6] > 0, 0]
    assert np.all(np.diff(acc) <= 1e-12 * acc[:-1] + 1e-15)
    assert np.all(poses[0] == s["poses0"][0])
    ate1 = O.pg_ate(s["poses_true"], poses)
    assert abs(ate1 - gold["ate_final"]) <= 1e-3 * gold["ate_final"] and ate1 < 0.05 * gold["ate_initial"]
    ps = e.pcg_summary()
    assert ps.hit_cap == 0 and ps.iterations_total <= 100 * summ.num_iterations and ps.coarse_dim == 942
    assert ps.coarse_failures == 0 and ps.coarse_refreshes >= summ.num_iterations      # every coarse operator factored (the flag is read now)
    # residual at the solution agrees with the oracle's evaluation of the same poses
    o = O.PG(poses, s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
    assert abs(o.evaluate(jac=False)[0] - summ.final_cost) <= 1e-9 * summ.final_cost


def test_pg_coarse_inverse_on_a_second_stream(st, O, c4):
    """round 6, stba_pcg_options::coarse_async = 1 (the default): the coarse operator is built and inverted on a second stream next to
    the PCG kernel and applied one LM iteration late -- except behind a long step, where the solve waits for its own inverse.  Ordered by
    events: two runs give the same bits.  Same LM iterations, final cost and (to 1e-5) converged poses as the in-line inverse; lagging
    ALWAYS (coarse_async_decrease = 1, or mode 2) is faster still and pays in the poses: a coarse solver that is stale by a long step
    leaves its error in the graph's weakly constrained modes.  coarse_eta -- a tolerance of its own on the coarse residual -- buys that back."""
    s, gold = c4
    gp = np.array(gold["final_poses_every_50th"]).reshape(-1, 7)
    res = {}
    for name, kw in (("inline", dict(coarse_async=0)), ("default", {}), ("default again", {}), ("always", dict(coarse_async_decrease=1.0)),
                     ("always tight", dict(coarse_async_decrease=1.0, coarse_eta=1e-4)), ("mode 2", dict(coarse_async=2))):
        e = st.PGEngine(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
        summ, tr, tot = e.solve(pcg=e.pcg_options(**kw))
        res[name] = (summ, tr, tot, e.get_poses(), e.pcg_summary())
        assert summ.termination_type == 0 and summ.num_iterations == gold["num_iterations"], name
        assert abs(summ.final_cost - gold["final_cost"]) <= 1e-9 * gold["final_cost"], name
        assert res[name][4].hit_cap == 0 and res[name][4].coarse_failures == 0 and res[name][4].one_kernel_solves == summ.num_iterations
    assert e.pcg_options().coarse_async == 1
    assert np.array_equal(res["default"][1], res["default again"][1]) and np.array_equal(res["default"][3], res["default again"][3])
    assert pose_diff(res["inline"][3][::50], gp) < 1e-5 and pose_diff(res["default"][3][::50], gp) < 1e-5
    assert abs(res["default"][2] - res["inline"][2]) <= 0.05 * res["inline"][2]            # (PCG iterations: 217 against 216)
    assert 1e-5 < pose_diff(res["always"][3][::50], gp) < 3e-4 and pose_diff(res["mode 2"][3][::50], gp) < 6e-4
    assert pose_diff(res["always tight"][3][::50], gp) < 5e-6


def test_pg_one_kernel_solve_and_the_way_back(st, O, c4):
    """round 5: the PCG solve of an LM iteration as one persistent kernel (two stamped exchanges per iteration).  Same LM trace as four
    launches per iteration (the PCG recurrences differ -- Chronopoulos-Gear -- so the bits do not have to agree, the iteration counts and
    the converged quantities do); the summary says which path ran; and the time-out path: a solve that gives up is repeated with
    launches and the result is the launches' own, bit for bit."""
    s, gold = c4
    res = {}
    for mode in (1, 0, 2, 11):                 # (11: the one-kernel solve once more -- run to run it must give the same bits)
        e = st.PGEngine(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
        summ, tr, pcg_total = e.solve(pcg=e.pcg_options(one_kernel_solve=mode % 10))
        res[mode] = (summ, tr, pcg_total, e.get_poses(), e.pcg_summary())
    assert np.array_equal(res[11][1], res[1][1]) and np.array_equal(res[11][3], res[1][3])
    assert res[1][4].one_kernel_solves == res[1][4].solves > 0 and res[0][4].one_kernel_solves == 0 and res[2][4].one_kernel_solves == 0
    # (inexact steps: a solve that stops one PCG iteration earlier or later takes a slightly different step, so the traces agree
    # loosely on the way and tightly where they converge)
    assert res[1][0].num_iterations == res[0][0].num_iterations and abs(res[1][2] - res[0][2]) <= 0.05 * res[0][2]
    assert np.allclose(res[1][1][:, 0], res[0][1][:, 0], rtol=1e-3) and abs(res[1][0].final_cost - res[0][0].final_cost) <= 1e-6 * res[0][0].final_cost
    assert pose_diff(res[1][3], res[0][3]) < 1e-5
    assert np.array_equal(res[2][1], res[0][1]) and np.array_equal(res[2][3], res[0][3]) and res[2][2] == res[0][2]
    # small graphs take the same path (3 groups of 8 nodes here) and follow the dense oracle
    import importlib
    scenes = importlib.import_module("slam-tricks_amd.scenes")
    g = scenes.pose_graph_scene(n_nodes=24, loops_per_node=2, seed=2, sigma_t=0.02, sigma_r=0.004, turns=2)
    e = st.PGEngine(g["poses0"], g["edge_i"], g["edge_j"], g["meas"], g["node_fixed"])
    summ, tr, _ = e.solve(pcg=e.pcg_options(forcing_eta0=0.0))
    so, tro = O.PG(g["poses0"], g["edge_i"], g["edge_j"], g["meas"], g["node_fixed"]).solve()
    assert e.pcg_summary().one_kernel_solves == summ.num_iterations == so.num_iterations
    assert np.allclose(tr[:, 0], tro[: len(tr), 0], rtol=1e-7)
