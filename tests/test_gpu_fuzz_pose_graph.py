"""Differential fuzz of the pose-graph path (config C4's engine) against the oracle: seeded random graph SHAPES -- 3 to 120 nodes
with 0 to 5 loop closures per node, shuffled edge order, reversed edges, repeated edges, extra fixed nodes, three noise levels.
Every case: residuals and Jacobians element by element, then the LM solve with exact steps (PCG to 1e-12) against the oracle's
DENSE LM iteration for iteration, with and without the coarse space; larger graphs (200-1500 nodes) with the production options
against the oracle's matrix-free LM on converged quantities (north_star: 1e-6 on the cost, 1e-5 on poses)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SMALL, MID = 20, 6


@pytest.fixture(scope="module")
def st():
    mod = importlib.import_module("slam-tricks_amd")
    assert mod.device_count() > 0
    return mod


def _case(scenes, k, mid=False):
    rng = np.random.default_rng((7000 if mid else 3000) + k)
    n = int(rng.integers(200, 1501)) if mid else int(rng.integers(3, 121))
    loops = int(rng.integers(1 if mid else 0, 6))          # (0: a pure odometry chain -- the integrated start is already the optimum)
    sig = [(0.01, 0.002), (0.02, 0.004), (0.04, 0.008)][int(rng.integers(0, 3))]
    s = scenes.pose_graph_scene(n_nodes=n, loops_per_node=loops, seed=int(rng.integers(1, 10000)), sigma_t=sig[0], sigma_r=sig[1],
                                turns=int(rng.integers(1, 9)))
    ei, ej, meas = s["edge_i"].copy(), s["edge_j"].copy(), s["meas"].copy()
    what = []
    if rng.random() < 0.4 and len(ei) > 4:                       # repeated edges (two measurements of one relative pose)
        extra = rng.choice(len(ei), max(1, len(ei) // 10), replace=False)
        ei, ej, meas = np.concatenate([ei, ei[extra]]), np.concatenate([ej, ej[extra]]), np.concatenate([meas, meas[extra]])
        what.append("repeated edges")
    if rng.random() < 0.7:                                       # the engine must not depend on the edge order
        p = rng.permutation(len(ei))
        ei, ej, meas = ei[p], ej[p], meas[p]
        what.append("shuffled")
    fixed = s["node_fixed"].copy()
    if rng.random() < 0.4 and n > 6:
        fixed[rng.choice(np.arange(1, n), int(rng.integers(1, 4)), replace=False)] = 1
        what.append("more fixed nodes")
    return dict(poses0=s["poses0"], edge_i=np.ascontiguousarray(ei, np.int32), edge_j=np.ascontiguousarray(ej, np.int32),
                meas=np.ascontiguousarray(meas), node_fixed=fixed, what=what, rng=rng)


def pose_diff(a, b):
    dq = np.minimum(np.abs(a[:, :4] - b[:, :4]).max(1), np.abs(a[:, :4] + b[:, :4]).max(1)).max()
    return max(dq, np.abs(a[:, 4:] - b[:, 4:]).max())


@pytest.mark.parametrize("k", range(SMALL))
def test_random_graph_exact_steps_follow_the_dense_oracle(st, O, scenes, k):
    c = _case(scenes, k)
    args = (c["poses0"], c["edge_i"], c["edge_j"], c["meas"], c["node_fixed"])
    e, o = st.PGEngine(*args), O.PG(*args)
    cost, r, Ji, Jj = e.evaluate()
    co, ro, Jio, Jjo = o.evaluate()
    assert abs(cost - co) <= 1e-12 * co + 1e-24          # (pure odometry chains start at a cost of 1e-28: rounding noise)
    assert np.abs(r - ro).max() < 1e-12 and np.abs(Ji - Jio).max() < 1e-11 and np.abs(Jj - Jjo).max() < 1e-11
    so, tro = o.solve()
    for group in (0, -1):
        e = st.PGEngine(*args)
        summ, tr, pcg_total = e.solve(pcg=e.pcg_options(forcing_eta0=0.0, coarse_group=group, max_iterations=4000))
        assert summ.termination_type == so.termination_type and summ.num_iterations == so.num_iterations, (group, c["what"])
        n = min(len(tr), len(tro))
        assert np.allclose(tr[:n, 0], tro[:n, 0], rtol=1e-7, atol=1e-14), (group, c["what"])
        assert np.all(tr[:n, 6] == tro[:n, 6])
        assert pose_diff(e.get_poses(), o.poses) < 1e-6
        ps = e.pcg_summary()
        assert ps.hit_cap == 0 and ps.coarse_failures == 0
        assert np.array_equal(e.get_poses()[c["node_fixed"] != 0], c["poses0"][c["node_fixed"] != 0])     # constant nodes stay put


@pytest.mark.parametrize("k", range(MID))
def test_random_graph_production_solve_reaches_the_oracles_answer(st, O, scenes, k):
    c = _case(scenes, k, mid=True)
    args = (c["poses0"], c["edge_i"], c["edge_j"], c["meas"], c["node_fixed"])
    e, o = st.PGEngine(*args), O.PG(*args)
    tight = dict(function_tolerance=1e-12, parameter_tolerance=1e-11)
    summ, tr, pcg_total = e.solve(**tight)
    so = o.solve_sparse(**tight)[0]
    assert summ.termination_type == 0 and so.termination_type == 0
    assert abs(summ.final_cost - so.final_cost) <= 1e-6 * max(so.final_cost, 1e-30), c["what"]
    assert pose_diff(e.get_poses(), o.poses) < 1e-5, c["what"]
    ps = e.pcg_summary()
    assert ps.hit_cap == 0 and ps.coarse_failures == 0
