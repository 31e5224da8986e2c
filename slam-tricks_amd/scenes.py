"""Seeded synthetic scenes for the NLS path (the reference's L3 "scene / data simulation" layer).

Every generator restates a reference scene with a fixed seed instead of the wall clock
(the reference seeds from the clock: scene.cpp:23, sim_data.cpp:273, two_view_simu.cpp:27):

  st20_scene      st20-g2o/src/src/sim_data.cpp:22-172,244-314   (BA: BASELINE configs C5, and the
                                                                  reference's own 29 x 600 size)
  two_view_scene  st22-two-view/src/src/two_view_simu.cpp:10-45   (BASELINE config C2)
  pnp_scene       st17-ceres/src/main.cpp:14-87, scene.cpp:11-43  (1-camera PnP)
  curve_fit_data  st7-ransac/src/include/parabola.hpp:29-43       (BASELINE config C1)
  calib_scene     st3-calibration (synthetic 20 x 88 board views) (BASELINE config C3)

Pure numpy; no dependency on oracle/ (this file is product-side input generation).
Camera pose = camera-to-world, quaternion (x, y, z, w) + position, 7 doubles per camera.
"""
import numpy as np


# --------------------------------------------------------------------------- small SO3 helpers
def quat_from_rot(R):
    R = np.asarray(R, dtype=np.float64)
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s, (R[2, 1] - R[1, 2]) / s]
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s, (R[0, 2] - R[2, 0]) / s]
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s, (R[1, 0] - R[0, 1]) / s]
    q = np.array(q)
    return q / np.linalg.norm(q)


def rot_from_quat(q):
    """(..., 4) -> (..., 3, 3)"""
    q = np.asarray(q, dtype=np.float64)
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z); R[..., 0, 1] = 2 * (x * y - w * z); R[..., 0, 2] = 2 * (x * z + w * y)
    R[..., 1, 0] = 2 * (x * y + w * z); R[..., 1, 1] = 1 - 2 * (x * x + z * z); R[..., 1, 2] = 2 * (y * z - w * x)
    R[..., 2, 0] = 2 * (x * z - w * y); R[..., 2, 1] = 2 * (y * z + w * x); R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def axis_angle(axis, ang):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


def project(cams, pts, obs_cam, obs_pt):
    """normalised image-plane projection of pts[obs_pt] in cams[obs_cam] (test_ceres.h:66-71)"""
    R = rot_from_quat(cams[obs_cam, :4])
    d = pts[obs_pt] - cams[obs_cam, 4:7]
    pc = np.einsum("nji,nj->ni", R, d)      # R^T d
    return pc[:, :2] / pc[:, 2:3], pc[:, 2]


def triangulate(cams, pts, obs_cam, obs_pt, obs_feat, iters=30):
    """Per-landmark refinement with cameras fixed (sim_data.h:165-194, sim_data.cpp:299-311),
    batched damped Gauss-Newton over all landmarks at once."""
    pts = pts.copy()
    n_pts = len(pts)
    R = rot_from_quat(cams[obs_cam, :4])
    Rt = np.transpose(R, (0, 2, 1))
    t = cams[obs_cam, 4:7]
    lam = np.full(n_pts, 1e-4)

    def resid(P):
        pc = np.einsum("nij,nj->ni", Rt, P[obs_pt] - t)
        return pc[:, :2] / pc[:, 2:3] - obs_feat, pc

    # (large scenes -- the landmark-heavy multi-GPU workload, 10^7 observations -- sum per landmark with reduceat over the
    # landmark-major segments: np.add.at takes minutes there.  Small scenes keep add.at: the frozen oracle traces under
    # tests/golden were made from its bits.)
    fast = len(obs_pt) > 2_000_000 and np.all(np.diff(obs_pt) >= 0)
    seg = np.flatnonzero(np.r_[True, np.diff(obs_pt) > 0]) if fast else None
    seg_ids = obs_pt[seg] if fast else None

    def scatter_sum(shape, vals):
        out = np.zeros(shape)
        if fast:
            out[seg_ids] = np.add.reduceat(vals, seg, axis=0)
        else:
            np.add.at(out, obs_pt, vals)
        return out

    def cost_of(r):
        return scatter_sum(n_pts, (r * r).sum(1))

    r, pc = resid(pts)
    cost = cost_of(r)
    for _ in range(iters):
        iz = 1.0 / pc[:, 2]
        A = np.zeros((len(obs_pt), 2, 3))
        A[:, 0, 0] = iz; A[:, 0, 2] = -pc[:, 0] * iz * iz
        A[:, 1, 1] = iz; A[:, 1, 2] = -pc[:, 1] * iz * iz
        J = A @ Rt
        H = scatter_sum((n_pts, 3, 3), np.einsum("nki,nkj->nij", J, J))
        g = scatter_sum((n_pts, 3), -np.einsum("nki,nk->ni", J, r))
        Hd = H.copy()
        idx = np.arange(3)
        Hd[:, idx, idx] += lam[:, None] * (H[:, idx, idx] + 1e-12)
        d = np.linalg.solve(Hd, g[..., None])[..., 0]
        cand = pts + d
        rn, pcn = resid(cand)
        cn = cost_of(rn)
        ok = (cn < cost) & np.isfinite(cn)
        pts[ok] = cand[ok]
        lam = np.where(ok, np.maximum(lam * 0.1, 1e-12), np.minimum(lam * 10, 1e12))
        cost = np.where(ok, cn, cost)
        r, pc = resid(pts)
        if np.max(np.abs(d[ok])) < 1e-13 if ok.any() else True:
            break
    return pts


# --------------------------------------------------------------------------- st20 BA scene
def spiral_cameras(n_cams, radius=3.0):
    """CreateTrajectory, sim_data.cpp:47-96: sphere spiral of radius 3 looking at the origin.
    The reference steps z by 0.02 and the azimuth by 10 deg and keeps every 10th sample
    (29 cameras); here the same curve is sampled n_cams times (identical for n_cams = 29)."""
    k = np.arange(n_cams)
    z = -radius + 0.1 + (2 * radius - 0.2) * k / n_cams
    deg = 2900.0 * k / n_cams
    rad = np.deg2rad(deg)
    cos_t = np.sqrt(radius * radius - z * z) / radius
    pos = np.stack([radius * cos_t * np.cos(rad), radius * cos_t * np.sin(rad), z], 1)
    cams = np.zeros((n_cams, 7))
    for i in range(n_cams):
        tr = pos[i]
        xa = np.array([-tr[1], tr[0], 0.0]); xa /= np.linalg.norm(xa)
        za = -tr / np.linalg.norm(tr)
        ya = np.cross(za, xa)
        R = np.stack([xa, ya, za], 1)
        cams[i, :4] = quat_from_rot(R)
        cams[i, 4:] = tr
    return cams


def cube_landmarks(n_pts, rng, half=5.0):
    """CreateScene, sim_data.cpp:22-45: features on the six faces of the 10 m cube; stored as
    float like pcl::PointXYZRGBA does."""
    face = np.arange(n_pts) % 6
    uv = rng.uniform(-half, half, size=(n_pts, 2))
    P = np.zeros((n_pts, 3))
    axis = face // 2
    sign = np.where(face % 2 == 0, 1.0, -1.0)
    for a in range(3):
        m = axis == a
        o = [i for i in range(3) if i != a]
        P[m, a] = sign[m] * half
        P[m, o[0]] = uv[m, 0]
        P[m, o[1]] = uv[m, 1]
    return P.astype(np.float32).astype(np.float64)


def st20_scene(n_cams=29, n_pts=600, max_obs_per_pt=None, seed=20, pos_noise=0.3, ang_noise_deg=3.0,
               pix_noise=0.0, half_w=0.8, half_h=0.6, retriangulate=True, chunk=4096):
    """ProblemScene + Simulation(true, 0.3, 3.0) (test_ceres.cpp:8-13).

    Returns dict with truth and noisy initial values, landmark-major observations and the
    6-dof fixed mask (first / last camera constant: sim_data.cpp:295-296, test_ceres.h:127-130).
    max_obs_per_pt = K keeps the K nearest in-view cameras per landmark."""
    rng = np.random.default_rng(seed)
    cams_true = spiral_cameras(n_cams)
    pts_true = cube_landmarks(n_pts, rng)
    R = rot_from_quat(cams_true[:, :4])           # (C,3,3)
    oc, op, of = [], [], []
    for s in range(0, n_pts, chunk):              # CreateMeasurements, sim_data.cpp:119-141
        P = pts_true[s:s + chunk]
        d = P[:, None, :] - cams_true[None, :, 4:7]            # (p,C,3)
        pc = np.einsum("cji,pcj->pci", R, d)
        z = pc[..., 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            xn, yn = pc[..., 0] / z, pc[..., 1] / z
        vis = (z > 0) & (np.abs(xn) < half_w) & (np.abs(yn) < half_h)
        if max_obs_per_pt is not None:
            dist = np.where(vis, np.linalg.norm(d, axis=2), np.inf)
            order = np.argsort(dist, axis=1, kind="stable")[:, :max_obs_per_pt]
            keep = np.zeros_like(vis)
            np.put_along_axis(keep, order, True, axis=1)
            vis &= keep
        pi, ci = np.nonzero(vis)                  # landmark-major, camera index ascending
        oc.append(ci); op.append(pi + s)
        of.append(np.stack([xn[pi, ci], yn[pi, ci]], 1))
    obs_cam = np.concatenate(oc).astype(np.int32)
    obs_pt = np.concatenate(op).astype(np.int64)
    obs_feat = np.concatenate(of)
    # landmarks seen by fewer than two cameras cannot be triangulated: drop them
    cnt = np.bincount(obs_pt, minlength=n_pts)
    good = cnt >= 2
    remap = np.cumsum(good) - 1
    m = good[obs_pt]
    obs_cam, obs_feat = obs_cam[m], obs_feat[m]
    obs_pt = remap[obs_pt[m]].astype(np.int32)
    pts_true = pts_true[good]
    # features are stored in pcl::PointXY (float): sim_data.cpp:136-137
    obs_feat = obs_feat.astype(np.float32).astype(np.float64)
    if pix_noise > 0:
        obs_feat = obs_feat + rng.normal(0.0, pix_noise, obs_feat.shape)

    cams0 = cams_true.copy()
    if pos_noise > 0 or ang_noise_deg > 0:        # Simulation, sim_data.cpp:273-293
        a = rng.normal(0.0, np.deg2rad(ang_noise_deg), size=(n_cams, 3))
        dp = rng.normal(0.0, pos_noise, size=(n_cams, 3))
        for i in range(n_cams):
            Rn = R[i] @ (axis_angle([0, 0, 1], a[i, 0]) @ axis_angle([0, 1, 0], a[i, 1]) @ axis_angle([1, 0, 0], a[i, 2]))
            cams0[i, :4] = quat_from_rot(Rn)
            cams0[i, 4:] = cams_true[i, 4:] + dp[i]
    cams0[0] = cams_true[0]; cams0[-1] = cams_true[-1]        # sim_data.cpp:295-296
    cam_fixed = np.zeros((n_cams, 6), dtype=np.uint8)
    cam_fixed[0] = 1; cam_fixed[-1] = 1                        # test_ceres.h:127-130
    pts0 = pts_true.copy()
    if retriangulate:
        pts0 = triangulate(cams0, pts0, obs_cam, obs_pt, obs_feat)
    return dict(cams_true=cams_true, pts_true=pts_true, cams0=cams0, pts0=pts0, obs_cam=obs_cam,
                obs_pt=obs_pt, obs_feat=obs_feat, cam_fixed=cam_fixed)


# --------------------------------------------------------------------------- st22 two-view scene
def two_view_scene(n_pts=5000, seed=22, pos_noise=0.3, ang_noise_deg=3.0, pix_noise=0.0):
    """two_view_simu.cpp:27-56 with K = (400,400,300,200), 600x400 (st22 main.cpp:12-16);
    landmarks U[0,10]^3 resampled until visible in both views.  Observations are converted to
    the normalised plane.  Gauge (build choice): camera 0 constant, camera 1's
    x-position (the baseline direction) constant."""
    rng = np.random.default_rng(seed)
    fx = fy = 400.0; cx, cy, W, Hh = 300.0, 200.0, 600, 400
    cams_true = np.zeros((2, 7))
    cams_true[0, :4] = quat_from_rot(np.eye(3)); cams_true[0, 4:] = [3.0, 5.0, 0.0]
    cams_true[1, :4] = quat_from_rot(axis_angle([0.0, -1.0, 0.0], np.pi / 4)); cams_true[1, 4:] = [7.0, 5.0, 0.0]
    R = rot_from_quat(cams_true[:, :4])
    pts = np.zeros((0, 3))
    while len(pts) < n_pts:
        cand = rng.uniform(0.0, 10.0, size=(4 * n_pts, 3))
        ok = np.ones(len(cand), dtype=bool)
        for c in range(2):
            pc = (cand - cams_true[c, 4:]) @ R[c]
            u = fx * pc[:, 0] / pc[:, 2] + cx
            v = fy * pc[:, 1] / pc[:, 2] + cy
            ok &= (pc[:, 2] > 0) & (u >= 0) & (u <= W - 1) & (v >= 0) & (v <= Hh - 1)
        pts = np.concatenate([pts, cand[ok]])[:n_pts]
    obs_pt = np.repeat(np.arange(n_pts, dtype=np.int32), 2)
    obs_cam = np.tile(np.array([0, 1], dtype=np.int32), n_pts)
    obs_feat, _ = project(cams_true, pts, obs_cam, obs_pt)
    if pix_noise > 0:
        obs_feat = obs_feat + rng.normal(0.0, pix_noise, obs_feat.shape)
    cams0 = cams_true.copy()
    a = rng.normal(0.0, np.deg2rad(ang_noise_deg), 3)
    Rn = R[1] @ (axis_angle([0, 0, 1], a[0]) @ axis_angle([0, 1, 0], a[1]) @ axis_angle([1, 0, 0], a[2]))
    cams0[1, :4] = quat_from_rot(Rn)
    dp = rng.normal(0.0, pos_noise, 3); dp[0] = 0.0
    cams0[1, 4:] += dp
    cam_fixed = np.zeros((2, 6), dtype=np.uint8)
    cam_fixed[0] = 1
    cam_fixed[1, 3] = 1
    pts0 = triangulate(cams0, pts, obs_cam, obs_pt, obs_feat)
    return dict(cams_true=cams_true, pts_true=pts, cams0=cams0, pts0=pts0, obs_cam=obs_cam, obs_pt=obs_pt,
                obs_feat=obs_feat, cam_fixed=cam_fixed)


def two_view_pairs(n_pts=5000, seed=22, pix_noise=0.0):
    """pixel correspondences of the st22 simulation (two_view_simu.cpp:27-56; same geometry as
    two_view_scene): returns dict(f1, f2 (n, 2) pixels, K, R_true, t_true = pose of frame 2 in frame 1,
    pts_f1 = landmarks in frame 1)"""
    s = two_view_scene(n_pts=n_pts, seed=seed, pix_noise=0.0)
    K = np.array([[400.0, 0, 300.0], [0, 400.0, 200.0], [0, 0, 1.0]])
    R = rot_from_quat(s["cams_true"][:, :4])
    pos = s["cams_true"][:, 4:]
    pw = s["pts_true"]
    out = {}
    for c in range(2):
        pc = (pw - pos[c]) @ R[c]
        out[c] = np.stack([K[0, 0] * pc[:, 0] / pc[:, 2] + K[0, 2], K[1, 1] * pc[:, 1] / pc[:, 2] + K[1, 2]], 1)
    rng = np.random.default_rng(seed + 1000)
    if pix_noise > 0:
        out[0] = out[0] + rng.normal(0.0, pix_noise, out[0].shape)
        out[1] = out[1] + rng.normal(0.0, pix_noise, out[1].shape)
    return dict(f1=out[0], f2=out[1], K=K, R_true=R[0].T @ R[1], t_true=R[0].T @ (pos[1] - pos[0]),
                pts_f1=(pw - pos[0]) @ R[0])


# --------------------------------------------------------------------------- st17 PnP scene
def _ypr_pose(yaw, pitch, roll):
    """CameraPose(), st17 main.cpp:14-35; DegreeToRadian is float (scene.h:36-39)."""
    d2r = np.float32(np.pi / 180.0)
    y = axis_angle([0, 0, 1], float(np.float32(d2r * np.float32(yaw))))
    p = axis_angle([1, 0, 0], float(np.float32(d2r * np.float32(pitch))))
    r = axis_angle([0, 1, 0], float(np.float32(d2r * np.float32(roll))))
    return (r @ p @ y).T                          # angleAxis.inverse().matrix()


def pnp_published_poses():
    """true / init camera-to-world poses of st17 (main.cpp:17-32), as published in
    st17-ceres/img/release.png: q_true = +-(0.40958, 0.70941, -0.49673, -0.28679), t = (3,2,1);
    q_init = (0.45452, 0.54168, -0.54168, -0.45452), t = (2.5, 0, 0)."""
    real = np.concatenate([quat_from_rot(_ypr_pose(-120.0, 110.0, 0.0)), [3.0, 2.0, 1.0]])
    init = np.concatenate([quat_from_rot(_ypr_pose(-90.0, 90.0, 10.0)), [2.5, 0.0, 0.0]])
    return real, init


def pnp_scene(seed=17, n_per_plane=10):
    """CubePlanes() + correspondence selection, st17 main.cpp:37-87, scene.cpp:11-43."""
    rng = np.random.default_rng(seed)
    real, init = pnp_published_poses()
    planes = [(0, 0, 0, -5, 0, 0, 10, 4.5), (0, 0, 90, 0, 5, 0, 10, 4.5), (0, 0, 0, 5, 0, 0, 10, 4.5),
              (0, 0, 90, 0, -5, 0, 10, 4.5), (90, 0, 0, 0, 0, -2.25, 10, 10)]
    pts = []
    for roll, pitch, yaw, dx, dy, dz, width, height in planes:
        Rp = (axis_angle([0, 1, 0], np.deg2rad(roll)) @ axis_angle([1, 0, 0], np.deg2rad(pitch))
              @ axis_angle([0, 0, 1], np.deg2rad(yaw)))
        uy = rng.uniform(-0.5 * width, 0.5 * width, n_per_plane)
        uz = rng.uniform(-0.5 * height, 0.5 * height, n_per_plane)
        local = np.stack([np.zeros(n_per_plane), uy, uz], 1)
        pts.append((local @ Rp.T + np.array([dx, dy, dz])).astype(np.float32).astype(np.float64))
    pts = np.concatenate(pts)
    Rr = rot_from_quat(real[:4])
    pc = (pts - real[4:]) @ Rr
    xn, yn = pc[:, 0] / pc[:, 2], pc[:, 1] / pc[:, 2]
    keep = (pc[:, 2] > 0) & (xn > -1) & (xn < 1) & (yn > -0.75) & (yn < 0.75)     # main.cpp:71-74
    return dict(pose_true=real, pose_init=init, pts=pts[keep], feats=np.stack([xn[keep], yn[keep]], 1))


# --------------------------------------------------------------------------- st7 curve fit (C1)
def curve_fit_data(n=1000, seed=17, abc=(1.0, 2.0, 3.0), sigma=0.1):
    """genData, parabola.hpp:29-43 (noise is added to x inside computeVal there), n points."""
    rng = np.random.default_rng(seed)
    a, b, c = abc
    mid = -b / (2 * a)
    x = rng.uniform(mid - 2.0, mid + 2.0, n)
    xe = x + rng.normal(0.0, sigma, n)
    y = a * xe * xe + b * xe + c
    return np.stack([x, y], 1)


# --------------------------------------------------------------------------- st3 calibration (C3)
def se3_exp(xi):
    """Sophus SE3::exp, tangent [rho, theta] -> (R, t)"""
    rho, th = np.asarray(xi[:3], float), np.asarray(xi[3:], float)
    a = np.linalg.norm(th)
    K = np.array([[0, -th[2], th[1]], [th[2], 0, -th[0]], [-th[1], th[0], 0]])
    if a < 1e-10:
        R = np.eye(3) + K + 0.5 * K @ K
        V = np.eye(3) + 0.5 * K + K @ K / 6.0
    else:
        R = np.eye(3) + np.sin(a) / a * K + (1 - np.cos(a)) / (a * a) * K @ K
        V = np.eye(3) + (1 - np.cos(a)) / (a * a) * K + (a - np.sin(a)) / (a ** 3) * K @ K
    return R, V @ rho


def se3_log(R, t):
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    a = np.arccos(c)
    if a < 1e-10:
        th = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2
    else:
        th = a / (2 * np.sin(a)) * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    K = np.array([[0, -th[2], th[1]], [th[2], 0, -th[0]], [-th[1], th[0], 0]])
    if a < 1e-10:
        V = np.eye(3) + 0.5 * K + K @ K / 6.0
    else:
        V = np.eye(3) + (1 - np.cos(a)) / (a * a) * K + (a - np.sin(a)) / (a ** 3) * K @ K
    return np.concatenate([np.linalg.solve(V, t), th])


def calib_forward(intr, xis, obj):
    """pixel predictions (V, C, 2) for params intr[9], xis (V,6), obj (V,C,2)  (calib.cpp:318-332)"""
    alpha, beta, u0, v0, k1, k2, k3, p1, p2 = intr
    out = np.zeros(obj.shape)
    for v in range(len(xis)):
        R, t = se3_exp(xis[v])
        P = obj[v] @ R[:, :2].T + t
        xn, yn = P[:, 0] / P[:, 2], P[:, 1] / P[:, 2]
        r2 = xn * xn + yn * yn
        rad = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
        xd = xn * rad + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn)
        yd = yn * rad + 2 * p2 * xn * yn + p1 * (r2 + 2 * yn * yn)
        out[v, :, 0] = alpha * xd + u0
        out[v, :, 1] = beta * yd + v0
    return out


def calib_scene(n_views=20, rows=8, cols=11, square=0.028, seed=3, pix_noise=0.3):
    """Synthetic C3: 20 views of an 8 x 11 board (88 corners), K ~ (3040, 3040, 2005, 1468),
    distortion ~ the real fixture's result, poses spread around the fixture's geometry."""
    rng = np.random.default_rng(seed)
    intr = np.array([3040.0, 3038.0, 2005.0, 1468.0, 0.2, -1.3, 2.4, 1e-4, -1e-3])
    jj, ii = np.meshgrid(np.arange(cols), np.arange(rows))
    board = np.stack([jj.reshape(-1) * square, ii.reshape(-1) * square], 1)     # calib.cpp:28
    obj = np.repeat(board[None], n_views, 0)
    xis = np.zeros((n_views, 6))
    for v in range(n_views):
        ang = rng.normal(0.0, 0.25, 3)
        Rv = axis_angle([1, 0, 0], ang[0]) @ axis_angle([0, 1, 0], ang[1]) @ axis_angle([0, 0, 1], ang[2])
        centre = np.array([0.5 * (cols - 1) * square, 0.5 * (rows - 1) * square, 0.0])
        t = np.array([rng.normal(0, 0.03), rng.normal(0, 0.03), 0.45 + rng.uniform(-0.08, 0.08)]) - Rv @ centre
        xis[v] = se3_log(Rv, t)
    img = calib_forward(intr, xis, obj) + rng.normal(0.0, pix_noise, obj.shape)
    return dict(intr_true=intr, xis_true=xis, obj=obj, img=img, rows=rows, cols=cols, square=square)


# --------------------------------------------------------------------------- pose graph (C4)
def _se3_mul(a, b):
    """compose 7-double poses (q xyzw, t): a * b, vectorised over the first axis"""
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    q = np.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                  aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz], -1)
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    t = np.einsum("...ij,...j->...i", rot_from_quat(a[..., :4]), b[..., 4:]) + a[..., 4:]
    return np.concatenate([q, t], -1)


def _se3_inv(a):
    q = a[..., :4] * np.array([-1.0, -1.0, -1.0, 1.0])
    t = -np.einsum("...ji,...j->...i", rot_from_quat(a[..., :4]), a[..., 4:])
    return np.concatenate([q, t], -1)


def _se3_exp7(xi):
    R, t = se3_exp(xi)
    return np.concatenate([quat_from_rot(R), t])


def pose_graph_scene(n_nodes=10000, loops_per_node=3, seed=4, sigma_t=0.01, sigma_r=0.002, radius=10.0, turns=10):
    """BASELINE config C4 (build-defined; the reference has no pose-graph code).  Nodes on the st4
    sphere spiral (st4-kalman/src/src/pose_simulation.cpp:21-66: `turns` revolutions while z climbs
    the sphere, body z-axis towards the sphere centre), n-1 odometry edges plus `loops_per_node`
    loop closures per node to nodes about one revolution away; measurements
    Z_ij = T_i^-1 T_j exp(noise); initial poses by integrating the noisy odometry; node 0 fixed."""
    rng = np.random.default_rng(seed)
    k = np.arange(n_nodes)
    theta = 2 * np.pi * turns * k / (n_nodes - 1)
    z = 0.02 * radius + 1.96 * radius * k / (n_nodes - 1)
    ri = np.sqrt(np.maximum(radius * radius - (z - radius) ** 2, 1e-12))
    pos = np.stack([ri * np.cos(theta), ri * np.sin(theta), z], 1)
    centre = np.array([0.0, 0.0, radius])
    poses = np.zeros((n_nodes, 7))
    for i in range(n_nodes):
        za = centre - pos[i]; za /= np.linalg.norm(za)
        xa = np.array([-np.sin(theta[i]), np.cos(theta[i]), 0.0])
        xa = xa - za * (xa @ za); xa /= np.linalg.norm(xa)
        ya = np.cross(za, xa)
        poses[i, :4] = quat_from_rot(np.stack([xa, ya, za], 1))
        poses[i, 4:] = pos[i]
    ei = list(range(n_nodes - 1)); ej = list(range(1, n_nodes))
    per_turn = max(2, (n_nodes - 1) // turns)
    for i in range(n_nodes):
        for _ in range(loops_per_node):
            jit = int(rng.integers(-max(1, per_turn // 50), max(2, per_turn // 50 + 1)))
            j = i + per_turn + jit
            if j >= n_nodes:                      # last revolution: close the loop backwards
                j = i - per_turn + jit
            if 0 <= j < n_nodes and j != i:
                ei.append(i); ej.append(j)
    ei = np.array(ei, np.int32); ej = np.array(ej, np.int32)
    rel = _se3_mul(_se3_inv(poses[ei]), poses[ej])
    noise = np.concatenate([rng.normal(0, sigma_t, (len(ei), 3)), rng.normal(0, sigma_r, (len(ei), 3))], 1)
    meas = np.stack([_se3_mul(rel[e][None], _se3_exp7(noise[e])[None])[0] for e in range(len(ei))])
    init = np.zeros_like(poses)
    init[0] = poses[0]
    for i in range(1, n_nodes):
        init[i] = _se3_mul(init[i - 1][None], meas[i - 1][None])[0]
    fixed = np.zeros(n_nodes, np.uint8); fixed[0] = 1
    return dict(poses_true=poses, poses0=init, edge_i=ei, edge_j=ej, meas=meas, node_fixed=fixed)
