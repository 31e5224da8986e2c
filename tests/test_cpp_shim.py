"""The C++ operator-API layer (include/stba/ceres.h) driven by a restatement of the reference's
own call sites (tests/cpp/test_ceres_shim.cpp).  CPU: it compiles against the header + links
libstba.so and fails loudly without a device.  GPU: results match the oracle / the published
st17 poses."""
import importlib
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "cpp", "test_ceres_shim.cpp")
PKG = os.path.join(ROOT, "slam-tricks_amd")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    st = importlib.import_module("slam-tricks_amd")
    if not os.path.exists(st.LIB_PATH):
        importlib.import_module("slam-tricks_amd.build").build()
    out = str(tmp_path_factory.mktemp("cpp") / "test_ceres_shim")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), SRC,
                           "-L", PKG, "-lstba", f"-Wl,-rpath,{PKG}", "-o", out])
    return out


def run(exe, *args):
    p = subprocess.run([exe, *args], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    out = {}
    for line in p.stdout.splitlines():
        k, _, v = line.partition(" ")
        out[k] = v
    return out


def write_scene(path, s):
    with open(path, "wb") as f:
        f.write(struct.pack("iii", len(s["cams0"]), len(s["pts0"]), len(s["obs_cam"])))
        f.write(np.ascontiguousarray(s["cams0"], np.float64).tobytes())
        f.write(np.ascontiguousarray(s["pts0"], np.float64).tobytes())
        f.write(np.ascontiguousarray(s["obs_cam"], np.int32).tobytes())
        f.write(np.ascontiguousarray(s["obs_pt"], np.int32).tobytes())
        f.write(np.ascontiguousarray(s["obs_feat"], np.float64).tobytes())
        f.write(np.ascontiguousarray(s["cam_fixed"][:, 0], np.uint8).tobytes())


def write_pnp(path, s):
    with open(path, "wb") as f:
        f.write(struct.pack("i", len(s["pts"])))
        f.write(np.ascontiguousarray(s["pose_true"], np.float64).tobytes())
        f.write(np.ascontiguousarray(s["pose_init"], np.float64).tobytes())
        f.write(np.ascontiguousarray(np.hstack([s["pts"], s["feats"]]), np.float64).tobytes())


def test_shim_compiles_and_fails_loudly_without_device(exe):
    st = importlib.import_module("slam-tricks_amd")
    out = run(exe)
    if st.device_count() == 0:
        assert "term 2" in out["bound_0"] and "no CPU fallback" in out["bound_0"]
        assert out["bound_0"].split()[1] == "0"          # parameter untouched
    else:
        assert abs(float(out["bound_0"].split()[1]) - 3.0) < 1e-8


def test_a_loss_function_is_refused_loudly(exe):
    """VERDICT r5 missing 6: Problem::AddResidualBlock(cost, loss, ...) with a non-null loss used to store it and solve UNWEIGHTED.
    Now Solve() refuses: FAILURE, the parameter untouched, the message names the reason (also on stderr).  No device needed."""
    p = subprocess.run([exe, "loss"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0
    toks = p.stdout.split()
    assert toks[0] == "loss" and float(toks[2]) == 0.5 and toks[4] == "2" and toks[6] == "1"
    assert "LossFunction" in p.stdout and "not implemented" in p.stdout and "LossFunction" in p.stderr
    # ownership (the same run): a cost function shared by two residual blocks is deleted once with the problem that owns it
    # (two cost functions -> two deletions), a problem that does not take ownership deletes nothing
    own = [l for l in p.stdout.splitlines() if l.startswith("ownership")][0].split()
    assert own[2] == "2" and own[4] == "0"


def test_reference_ba_functor_is_recognised_on_the_host(exe, tmp_path, scenes):
    """Solve()'s dispatch (no device needed): the reference's own ProjectFactor behind DynamicAutoDiffCostFunction
    (test_ceres.h:47-81,109-121) is recognised as the reprojection factor, every `feature` is recovered exactly
    enough (1e-15), the observation structure is rebuilt; a functor with another residual is rejected."""
    s = scenes.st20_scene()
    f = str(tmp_path / "s.bin")
    write_scene(f, s)
    out = run(exe, "probe", f)
    toks = out["probe_user"].split()
    assert toks[:8] == ["detected", "1", "cams", str(len(s["cams0"])), "pts", str(len(s["pts0"])), "obs", str(len(s["obs_cam"]))]
    assert float(toks[-1]) < 1e-15
    assert out["probe_scaled"].split()[1] == "0"
    # a functor that IS the reprojection factor at every generic probe point and differs where (some of) its own data lie -- landmarks
    # further than `far` from the origin, `far` beyond every probe point: caught by the check at the blocks' own parameter values
    norms = np.linalg.norm(s["pts0"], axis=1)
    far = float(max(3.6, np.quantile(norms, 0.7)))
    assert (norms > far).any()
    out = run(exe, "probe", f, repr(far))
    assert out["probe_far"].split()[1] == "0" and out["probe_far_never"].split()[1] == "1"


def vec(out, key):
    return np.array([float(x) for x in out[key].split()])


def qerr(a, b):
    return min(np.abs(a - b).max(), np.abs(a + b).max())


@pytest.mark.gpu
def test_reference_call_sites_on_gpu(exe, tmp_path, scenes, O, known):
    pnp = scenes.pnp_scene(seed=17)
    sc = scenes.st20_scene()                                              # 29 x 600, the reference's size
    small = scenes.st20_scene(n_cams=10, n_pts=120, seed=3, pos_noise=0.1, ang_noise_deg=1.0)
    f_pnp, f_sc, f_small = (str(tmp_path / n) for n in ("pnp.bin", "scene.bin", "small.bin"))
    write_pnp(f_pnp, pnp); write_scene(f_sc, sc); write_scene(f_small, small)
    out = run(exe, f_pnp, f_sc, f_sc)
    # ceres_bound.cpp
    assert abs(float(out["bound_0"].split()[1]) - known["st17_ceres_bound"]["x_free"]) < 1e-8
    assert abs(float(out["bound_1"].split()[1]) - known["st17_ceres_bound"]["x_bounded"]) < 1e-12
    # PnP four ways -> published truth (release.png); wrong reference Jacobian needs more iterations
    truth = pnp["pose_true"]
    iters = {}
    for tag in ("pnp_dyn", "pnp_auto", "pnp_sized_0", "pnp_sized_1"):
        pose = vec(out, tag + "_pose")
        assert qerr(pose[:4], truth[:4]) < 1e-7 and np.abs(pose[4:] - truth[4:]).max() < 1e-6, tag
        toks = out[tag].split()
        iters[tag] = int(toks[toks.index("iters") + 1])
        assert float(toks[toks.index("final") + 1]) < 1e-14
    assert iters["pnp_dyn"] == iters["pnp_auto"]
    assert iters["pnp_sized_1"] > iters["pnp_sized_0"]                    # SURVEY fact 2 / release.png 8 vs 6
    toks = out["pnp_dyn"].split()
    assert int(toks[toks.index("callbacks") + 1]) == iters["pnp_dyn"]
    first, last = float(toks[toks.index("first_cb_x") + 1]), float(toks[toks.index("last_cb_x") + 1])
    assert abs(first - pnp["pose_init"][4]) > 1e-3 and abs(last - truth[4]) < 1e-6   # live state in callbacks
    # BA, built-in factor: device-resident path, matches the oracle
    assert out["ba_builtin_path"] == "gpu-ba"
    o = O.BA(sc["cams0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_feat"], sc["cam_fixed"])
    so, _ = o.solve()
    cams = vec(out, "ba_builtin_cams").reshape(-1, 7)
    toks = out["ba_builtin_term"].split()
    assert int(toks[0]) == 0 and int(toks[2]) == so.num_iterations
    assert abs(float(toks[6]) - so.final_cost) <= 1e-6 * max(so.final_cost, 1e-12) + 1e-15
    dq = np.minimum(np.abs(cams[:, :4] - o.cams[:, :4]).max(1), np.abs(cams[:, :4] + o.cams[:, :4]).max(1)).max()
    assert dq < 1e-8 and np.abs(cams[:, 4:] - o.cams[:, 4:]).max() < 1e-8
    assert np.all(cams[0] == sc["cams0"][0]) and np.all(cams[-1] == sc["cams0"][-1])
    # BA, the reference's UNCHANGED call site (user ProjectFactor behind DynamicAutoDiffCostFunction,
    # test_ceres.h:109-130): recognised numerically, runs on the device-resident engine, bit-identical to
    # the built-in factor and equal to the oracle's trace
    assert out["ba_user_path"] == "gpu-ba"
    tu, tb = out["ba_user_term"].split(), out["ba_builtin_term"].split()
    assert tu[:5] == tb[:5]                       # termination type, iteration count, initial cost (bit-equal features)
    # (the Schur kernel's LDS atomics make the last bits run-dependent: compare, do not expect bit equality)
    assert abs(float(tu[6]) - float(tb[6])) <= 1e-6 * float(tb[6]) + 1e-15
    assert np.abs(vec(out, "ba_user_cams") - vec(out, "ba_builtin_cams")).max() < 1e-9
    assert np.abs(vec(out, "ba_user_pts") - vec(out, "ba_builtin_pts")).max() < 1e-9
    cu = vec(out, "ba_user_cams").reshape(-1, 7)
    dq = np.minimum(np.abs(cu[:, :4] - o.cams[:, :4]).max(1), np.abs(cu[:, :4] + o.cams[:, :4]).max(1)).max()
    assert dq < 1e-8 and np.abs(cu[:, 4:] - o.cams[:, 4:]).max() < 1e-8
    # BA with a user functor that is NOT the reprojection factor (residual scaled by 2: rejected by the probe): BA-shaped, so it
    # runs on the device engine with the user's cost functions evaluated on the host in bulk -- "gpu-ba-hostjac" -- at the
    # reference's own size, and follows the oracle's trace (costs x 4: a scaled residual with Jacobi scaling is the same LM)
    assert out["ba_generic_path"] == "gpu-ba-hostjac"
    tg = out["ba_generic_term"].split()
    assert int(tg[0]) == 0 and int(tg[2]) == so.num_iterations
    _, tro = O.BA(sc["cams0"], sc["pts0"], sc["obs_cam"], sc["obs_pt"], sc["obs_feat"], sc["cam_fixed"]).solve()
    costs = vec(out, "ba_generic_costs")
    assert len(costs) == len(tro) and np.allclose(costs, 4.0 * tro[:, 0], rtol=1e-6, atol=1e-12)
    cams2 = vec(out, "ba_generic_cams").reshape(-1, 7)
    dq = np.minimum(np.abs(cams2[:, :4] - o.cams[:, :4]).max(1), np.abs(cams2[:, :4] + o.cams[:, :4]).max(1)).max()
    assert dq < 1e-8 and np.abs(cams2[:, 4:] - o.cams[:, 4:]).max() < 1e-8
    assert small is not None


@pytest.mark.gpu
def test_generic_ba_factor_beyond_the_dense_limit(exe, tmp_path, scenes, O):
    """100 cameras x 10 000 landmarks with a user factor the probe rejects: 600 + 30 000 local parameters, far beyond the
    4096 of the dense callback path (which used to refuse it).  Host-linearised device engine, the oracle's trace (x 4)."""
    s = scenes.st20_scene(n_cams=100, n_pts=10000, max_obs_per_pt=8, seed=31, pix_noise=1e-3)
    f = str(tmp_path / "g.bin")
    write_scene(f, s)
    out = run(exe, "generic_big", f, "4")
    assert out["ba_generic_path"] == "gpu-ba-hostjac", out["ba_generic_report"]
    o = O.BA(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
    so, tro = o.solve(max_num_iterations=4, num_threads=8)
    costs = vec(out, "ba_generic_costs")
    assert len(costs) == so.num_iterations + 1
    assert np.allclose(costs, 4.0 * tro[: len(costs), 0], rtol=1e-6)
    cams = vec(out, "ba_generic_cams").reshape(-1, 7)
    dq = np.minimum(np.abs(cams[:, :4] - o.cams[:, :4]).max(1), np.abs(cams[:, :4] + o.cams[:, :4]).max(1)).max()
    assert dq < 1e-6 and np.abs(cams[:, 4:] - o.cams[:, 4:]).max() < 1e-6


@pytest.mark.gpu
def test_published_pnp_workload_through_the_operator_api(exe, tmp_path, scenes):
    """round 6: the three st17 PnP call sites as bench.py times them (construction + Solve per call, solver.hpp:253-288): every
    repetition ends at the published pose, through the small dense path (one kernel launch per LM step), in well under the 3.5 ms
    a Solve() of this size took until round 5 -- the bound here is loose on purpose (a shared box), bench.py reports the number."""
    pnp = scenes.pnp_scene(seed=17)
    f = str(tmp_path / "pnp.bin")
    write_pnp(f, pnp)
    out = run(exe, "time_pnp", f, "20")
    truth = pnp["pose_true"]
    for tag in ("pnp_dyn", "pnp_auto", "pnp_sized"):
        toks = out[tag].split()
        d = {toks[i]: toks[i + 1] for i in range(0, len(toks), 2)}
        assert d["path"] == "gpu-dense-callback" and d["term"] == "0" and float(d["final"]) < 1e-14
        pose = vec(out, tag + "_pose")
        assert qerr(pose[:4], truth[:4]) < 1e-7 and np.abs(pose[4:] - truth[4:]).max() < 1e-6
        ms = vec(out, tag + "_ms")
        assert len(ms) == 20 and np.median(ms[1:]) < 2.0, ms


@pytest.mark.gpu
def test_drop_in_phases_are_reported(exe, tmp_path, scenes):
    """Solver::Summary carries Ceres' timing fields and the header's own phase timers; they add up to the wall time of Solve()"""
    s = scenes.st20_scene()
    f = str(tmp_path / "s.bin")
    write_scene(f, s)
    out = run(exe, "time_ba", f, "50", "2")
    toks = out["time_ba_1"].split()
    d = {toks[i]: toks[i + 1] for i in range(0, len(toks), 2)}
    assert d["path"] == "gpu-ba" and d["term"] == "0"
    parts = sum(float(d[k]) for k in ("recognise", "pack", "engine_create", "device_solve", "write_back", "verify", "resolve"))
    assert 0.0 < float(d["device_solve"]) <= parts <= float(d["solve"]) * 1.001 + 1e-4
    assert parts >= 0.9 * float(d["solve"]) - 1e-3
    assert float(d["minimizer"]) <= float(d["device_solve"]) + 1e-6


@pytest.mark.gpu
def test_reference_ba_call_site_at_c5_size(exe, tmp_path, scenes, O):
    """The reference's unchanged Ceres BA call site (test_ceres.h:98-152, one DynamicAutoDiffCostFunction per
    observation) at BASELINE config C5: 10^6 user cost functions are probed, the problem runs on the
    device-resident engine ("gpu-ba"; the callback path would refuse it), and three LM iterations match
    the oracle's cost trace to 1e-6 and its camera poses to 1e-5."""
    s = scenes.st20_scene(n_cams=1000, n_pts=100000, max_obs_per_pt=10, seed=20, pix_noise=1e-3)
    f = str(tmp_path / "c5.bin")
    write_scene(f, s)
    out = run(exe, "big", f, "3")
    assert out["ba_user_path"] == "gpu-ba", out["ba_user_report"]
    o = O.BA(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
    so, tro = o.solve(max_num_iterations=3, num_threads=16)
    costs = vec(out, "ba_user_costs")
    assert len(costs) == so.num_iterations + 1
    assert np.allclose(costs, tro[: len(costs), 0], rtol=1e-6)
    cams = vec(out, "ba_user_cams").reshape(-1, 7)
    dq = np.minimum(np.abs(cams[:, :4] - o.cams[:, :4]).max(1), np.abs(cams[:, :4] + o.cams[:, :4]).max(1)).max()
    assert dq < 1e-5 and np.abs(cams[:, 4:] - o.cams[:, 4:]).max() < 1e-5


@pytest.mark.gpu
def test_small_solves_end_with_the_same_bits_every_run(exe, tmp_path, scenes):
    """600 tiny Solve() calls (the reference's per-landmark triangulation, sim_data.cpp:298-311) through the one-launch-per-step
    path, five times over: every run prints the same landmarks, bit for bit.  Until the hand-off became a stamped block
    (common.hpp) a step's result was now and then read before it had arrived -- one run in thirteen ended a few landmarks
    somewhere else (tools/dbg/tri_repeat.py runs hundreds)."""
    s = scenes.st20_scene(retriangulate=False)
    f = str(tmp_path / "tri.bin")
    write_scene(f, s)
    lines = {run(exe, "tri", f)["tri_pts"] for _ in range(5)}
    assert len(lines) == 1


@pytest.mark.gpu
def test_reference_triangulation_call_site_on_gpu(exe, tmp_path, scenes, O):
    """sim_data.cpp:298-311: the scene generator triangulates every landmark with its OWN ceres::Problem (cameras are
    not parameter blocks; AutoDiffCostFunction<Triangulation, 2, 3> per observation, sim_data.h:165-194, residual
    feature - proj).  The restated call site runs at the reference's size (29 cameras, 600 landmarks) through the
    generic callback path of the shim; the landmarks must equal the bulk kernel's (stba_ba_triangulate, one landmark
    per lane) and the oracle's up to the stopping rule, and the oracle's dense LM on the same 600 problems tightly.  The wall time of the 600 tiny
    solves is printed next to the bulk kernel's: INTEGRATION.md 2 quotes the ratio."""
    import time
    st = importlib.import_module("slam-tricks_amd")
    s = scenes.st20_scene(retriangulate=False)                          # landmarks NOT triangulated yet
    f = str(tmp_path / "tri.bin")
    write_scene(f, s)
    out = run(exe, "tri", f)
    toks = out["tri_summary"].split()
    n_prob, n_conv, secs = int(toks[1]), int(toks[3]), float(toks[7])
    assert toks[9] == "gpu-dense-callback"
    n_seen = len(np.unique(s["obs_pt"]))
    assert n_prob == n_seen and n_conv == n_prob
    pts = vec(out, "tri_pts").reshape(-1, 3)
    e = st.BAEngine(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
    e.triangulate()                                                     # (warm-up: module load, allocations)
    e.set_params(s["cams0"], s["pts0"])
    t0 = time.perf_counter()
    e.triangulate()
    _, pts_bulk = e.get_params()
    t_bulk = time.perf_counter() - t0
    seen = np.zeros(len(pts), bool); seen[s["obs_pt"]] = True
    assert np.array_equal(pts[~seen], s["pts0"][~seen])
    # (a) the same 600 problems through the oracle's dense LM with the same (default) options: the same iterates, tightly
    cams = s["cams0"]

    def rot(q):
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    Rs = np.stack([rot(c[:4]) for c in cams])
    order = np.argsort(s["obs_pt"], kind="stable")
    starts = np.searchsorted(s["obs_pt"][order], np.arange(len(pts) + 1))
    worst, cost_shim, cost_bulk = 0.0, 0.0, 0.0

    def make(j):
        k = order[starts[j]:starts[j + 1]]
        Rk, tk, fk = Rs[s["obs_cam"][k]], cams[s["obs_cam"][k], 4:], s["obs_feat"][k]

        def res(p):
            pc = np.einsum("kji,kj->ki", Rk, p[None, :] - tk)               # R^T (p - t)
            iz = 1.0 / pc[:, 2]
            r = (fk - pc[:, :2] * iz[:, None]).reshape(-1)                   # feature - proj (sim_data.h:191)
            A = np.zeros((len(k), 2, 3))
            A[:, 0, 0] = iz; A[:, 0, 2] = -pc[:, 0] * iz * iz; A[:, 1, 1] = iz; A[:, 1, 2] = -pc[:, 1] * iz * iz
            J = -np.einsum("kab,kcb->kac", A, Rk).reshape(-1, 3)             # -A R^T
            return r, J
        return res, 2 * len(k)
    for j in np.nonzero(seen)[0]:
        res, nres = make(j)
        po, so, _ = O.dense_lm(res, s["pts0"][j], nres)
        worst = max(worst, float(np.abs(po - pts[j]).max()))
        cost_shim += 0.5 * float(np.sum(res(pts[j])[0] ** 2))
        cost_bulk += 0.5 * float(np.sum(res(pts_bulk[j])[0] ** 2))
    assert worst < 1e-8, worst
    # (b) against the bulk kernel and the oracle's triangulation, which iterate further than Ceres' default function
    # tolerance 1e-6 lets the call site go: same minimiser (weakly observed depths move by ~1e-4 on a 10 m scene between the
    # two stopping rules), the summed cost of the call site's landmarks within 1e-5 relative above the bulk result's
    o = O.BA(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
    o.triangulate()
    assert np.abs(pts_bulk[seen] - o.pts[seen]).max() < 1e-6
    assert np.abs(pts[seen] - pts_bulk[seen]).max() < 2e-3
    assert cost_bulk <= cost_shim * (1 + 1e-12) and cost_shim - cost_bulk <= 1e-5 * max(cost_bulk, 1e-30) + 1e-12, (cost_shim, cost_bulk)
    print(f"triangulation call site: {n_prob} per-landmark Solve() calls {secs * 1e3:.1f} ms "
          f"({secs / n_prob * 1e6:.0f} us each), bulk stba_ba_triangulate {t_bulk * 1e3:.2f} ms incl. read-back: ratio {secs / t_bulk:.0f}x")


# ---------------------------------------------------------------------------------------------
# g2o front door: st20-g2o/src/include/test_g2o.h restated on include/stba/g2o.h
@pytest.fixture(scope="module")
def exe_g2o(tmp_path_factory):
    st = importlib.import_module("slam-tricks_amd")
    if not os.path.exists(st.LIB_PATH):
        importlib.import_module("slam-tricks_amd.build").build()
    out = str(tmp_path_factory.mktemp("cpp") / "test_g2o_shim")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_g2o_shim.cpp"), "-L", PKG, "-lstba",
                           f"-Wl,-rpath,{PKG}", "-o", out])
    return out


def test_g2o_shim_compiles_and_fails_loudly_without_device(exe_g2o, tmp_path, scenes):
    st = importlib.import_module("slam-tricks_amd")
    s = scenes.st20_scene(n_cams=6, n_pts=40, seed=7, pos_noise=0.05, ang_noise_deg=1.0)
    f = str(tmp_path / "s.bin")
    write_scene(f, s)
    out = run(exe_g2o, f)
    if st.device_count() == 0:
        assert out["g2o_iters"].startswith("0 ") and "no CPU fallback" in out["g2o_iters"]


def test_g2o_probe_rejects_a_graph_with_one_foreign_edge(exe_g2o, tmp_path, scenes):
    """the front door checks EVERY edge's computeError (host work, before any device call): one edge of another
    type among the projection edges and optimize() refuses, naming the edge"""
    s = scenes.st20_scene(n_cams=6, n_pts=40, seed=7, pos_noise=0.05, ang_noise_deg=1.0)
    f = str(tmp_path / "s.bin")
    write_scene(f, s)
    out = run(exe_g2o, f, "odd")
    assert out["g2o_iters"].startswith("0 ")
    assert f"edge {len(s['obs_cam']) // 2}: computeError is not the reprojection residual" in out["g2o_iters"]


@pytest.mark.gpu
def test_solve_with_g2o_on_gpu(exe_g2o, tmp_path, scenes, O):
    """SolveWithG2O (test_g2o.h:94-147): no fixed vertex (gauge held by the LM damping), optimize(40),
    chi2 = 2 x cost.  Noise-free observations: the residual must vanish."""
    s = scenes.st20_scene()
    f = str(tmp_path / "s.bin")
    write_scene(f, s)
    out = run(exe_g2o, f)
    toks = out["g2o_iters"].split()
    iters, chi2 = int(toks[0]), float(toks[2])
    assert 1 <= iters <= 40, out["g2o_iters"]
    o = O.BA(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"])      # no constant cameras
    so, _ = o.solve(max_num_iterations=40)
    assert iters == so.num_iterations
    assert chi2 < 1e-10 and abs(chi2 - 2 * so.final_cost) < 1e-12
    cams = vec(out, "g2o_cams").reshape(-1, 7)
    dq = np.minimum(np.abs(cams[:, :4] - o.cams[:, :4]).max(1), np.abs(cams[:, :4] + o.cams[:, :4]).max(1)).max()
    assert dq < 1e-7 and np.abs(cams[:, 4:] - o.cams[:, 4:]).max() < 1e-6
    # with the gauge fixed like the Ceres caller does, the truth is recovered
    out = run(exe_g2o, f, "fix")
    cams = vec(out, "g2o_cams").reshape(-1, 7)
    ct = s["cams_true"]
    dq = np.minimum(np.abs(cams[:, :4] - ct[:, :4]).max(1), np.abs(cams[:, :4] + ct[:, :4]).max(1)).max()
    assert dq < 1e-6 and np.abs(cams[:, 4:] - ct[:, 4:]).max() < 1e-5


def write_pg(path, s):
    with open(path, "wb") as f:
        f.write(struct.pack("ii", len(s["poses0"]), len(s["edge_i"])))
        f.write(np.ascontiguousarray(s["poses0"], np.float64).tobytes())
        f.write(np.ascontiguousarray(s["edge_i"], np.int32).tobytes())
        f.write(np.ascontiguousarray(s["edge_j"], np.int32).tobytes())
        f.write(np.ascontiguousarray(s["meas"], np.float64).tobytes())
        f.write(np.ascontiguousarray(s["node_fixed"], np.uint8).tobytes())


@pytest.mark.gpu
def test_pose_graph_through_the_operator_api(exe, tmp_path, scenes, O):
    """A pose graph written against the Ceres-style API (one 7-double block per pose with the SE3 right-plus chart, one
    RelativePoseFactor per edge) is dispatched to the device pose-graph engine ("gpu-pg") and reaches the oracle's answer; the
    SAME problem forced onto the generic host path (user-visible Evaluate + ComputeJacobian, dense normal equations on the device)
    follows the oracle's exact-step trace -- which checks the factor's autodiff Jacobian and the chart against the oracle's
    closed forms."""
    s = scenes.pose_graph_scene(n_nodes=400, loops_per_node=3, seed=4, sigma_t=0.02, sigma_r=0.004, turns=6)
    f = str(tmp_path / "pg.bin")
    write_pg(f, s)
    o = O.PG(s["poses0"], s["edge_i"], s["edge_j"], s["meas"], s["node_fixed"])
    so, tro, _, _ = o.solve_sparse()
    out = run(exe, "pg", f)
    assert out["pg_path"] == "gpu-pg", out.get("pg_msg")
    toks = out["pg_term"].split()
    assert int(toks[0]) == 0 and abs(float(toks[4]) - so.initial_cost) <= 1e-12 * so.initial_cost
    assert abs(float(toks[6]) - so.final_cost) <= 1e-6 * so.final_cost
    poses = vec(out, "pg_poses").reshape(-1, 7)
    dq = np.minimum(np.abs(poses[:, :4] - o.poses[:, :4]).max(1), np.abs(poses[:, :4] + o.poses[:, :4]).max(1)).max()
    assert dq < 1e-5 and np.abs(poses[:, 4:] - o.poses[:, 4:]).max() < 1e-5
    assert np.all(poses[0] == s["poses0"][0])
    # generic path on a smaller graph (900 local parameters): exact steps, the oracle's trace
    s2 = scenes.pose_graph_scene(n_nodes=150, loops_per_node=3, seed=4, sigma_t=0.02, sigma_r=0.004, turns=6)
    f2 = str(tmp_path / "pg2.bin")
    write_pg(f2, s2)
    o2 = O.PG(s2["poses0"], s2["edge_i"], s2["edge_j"], s2["meas"], s2["node_fixed"])
    so2, tro2 = o2.solve()
    out2 = run(exe, "pg", f2, "dense")
    assert out2["pg_path"] == "gpu-dense-callback"
    toks = out2["pg_term"].split()
    assert int(toks[0]) == 0 and int(toks[2]) == so2.num_iterations
    assert np.allclose(vec(out2, "pg_costs"), tro2[:, 0], rtol=1e-7)
    p2 = vec(out2, "pg_poses").reshape(-1, 7)
    dq = np.minimum(np.abs(p2[:, :4] - o2.poses[:, :4]).max(1), np.abs(p2[:, :4] + o2.poses[:, :4]).max(1)).max()
    assert dq < 1e-7 and np.abs(p2[:, 4:] - o2.poses[:, 4:]).max() < 1e-6


@pytest.mark.gpu
def test_recognised_blocks_are_checked_at_their_data_before_and_after_the_solve(exe, tmp_path, scenes):
    """the recognition of a user cost function as the reprojection factor is no longer a matter of four generic probe
    points.  A functor that differs from the factor only where its own data lie is caught before the solve and runs its own code
    (gpu-ba-hostjac); one that departs from it only once the landmarks have moved is caught by the check at the END of the solve --
    the parameters go back, the problem is solved again with the user's Evaluate, and the summary says so; the reference's own functor
    passes both checks and the summary's message says how many blocks were taken over."""
    s = scenes.st20_scene(n_cams=10, n_pts=120, seed=3, pos_noise=0.05, ang_noise_deg=0.5)
    f = str(tmp_path / "s.bin")
    write_scene(f, s)
    norms = np.linalg.norm(s["pts0"], axis=1)
    far = float(max(3.6, np.quantile(norms, 0.7)))
    out = run(exe, "verify", f, repr(far))
    assert out["ba_far_path"] == "gpu-ba-hostjac" and out["ba_far_term"].split()[0] == "0"
    assert out["ba_moved_path"] == "gpu-ba-hostjac" and "differs from it at the solution" in out["ba_moved_report"]
    assert out["ba_moved_term"].split()[0] == "0"
    assert out["ba_user_path"] == "gpu-ba" and "recognised as the reprojection factor" in out["ba_user_report"]
