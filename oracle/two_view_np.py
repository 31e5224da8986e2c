"""ORACLE (test infrastructure, not a product path): numpy restatement of the reference's two-view
initialiser, st22-two-view/src/src/two_view_geometry.cpp.  Only tests/ may import this.

  fundamental()   ComputeFunctionMatrix  :18-41   (n x 9 system, last right singular vector, row-major F)
  decompose()     DecomposeFMat          :43-81   (E = K^T F K, four hypotheses, all-points cheirality)
  triangulate()   Triangulate            :105-126 (6 x 4 DLT)
  adjust()        AdjustRotationMatrix   two_view_simu.h:19-26
Pinned by: the st22 simulation itself (two_view_simu.cpp:27-56, noise-free): the recovered pose must equal
the simulated one up to the baseline length, and the triangulated points the simulated points (main.cpp:21-37).
"""
import numpy as np


def hat(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def fundamental(f1, f2):
    u1, v1, u2, v2 = f1[:, 0], f1[:, 1], f2[:, 0], f2[:, 1]
    A = np.stack([u1 * u2, u1 * v2, u1, v1 * u2, v1 * v2, v1, u2, v2, np.ones(len(u1))], 1)     # :24-32
    _, _, Vt = np.linalg.svd(A)
    return Vt[-1].reshape(3, 3)                                                                    # :36-38


def projection(K, R, t):
    return K @ np.hstack([R.T, (-R.T @ t)[:, None]])                                               # :108-116


def triangulate(x1, x2, R1, t1, R2, t2, K):
    P1, P2 = projection(K, R1, t1), projection(K, R2, t2)
    A = np.vstack([hat([x1[0], x1[1], 1.0]) @ P1, hat([x2[0], x2[1], 1.0]) @ P2])                  # :118-120
    _, _, Vt = np.linalg.svd(A)
    lm = Vt[-1]
    return lm[:3] / lm[3]                                                                          # :122-125


def adjust(R):
    U, _, Vt = np.linalg.svd(R)
    return U @ Vt


def check(R, t, K, f1, f2):
    """CheckRotMatTransVec :83-103: number of correspondences that FAIL (0 = hypothesis accepted)"""
    bad = 0
    for a, b in zip(f1, f2):
        p1 = triangulate(a, b, np.eye(3), np.zeros(3), R, t, K)
        p2 = R.T @ p1 - R.T @ t
        bad += not (p1[2] > 0.0 and p2[2] > 0.0)
    return bad


def decompose(F, K, f1, f2):
    E = K.T @ F @ K
    U, _, Vt = np.linalg.svd(E)
    W = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])
    t1, t2 = U[:, 2], -U[:, 2]
    R1, R2 = U @ W @ Vt, U @ W.T @ Vt
    if np.linalg.det(R1) < 0:
        R1 = -R1
    if np.linalg.det(R2) < 0:
        R2 = -R2
    hyps = [(R1, t1), (R1, t2), (R2, t1), (R2, t2)]                                                # :61-64
    fails = np.array([check(R, t, K, f1, f2) for R, t in hyps])
    ok = np.flatnonzero(fails == 0)
    if len(ok) != 1:
        return None, fails
    R, t = hyps[ok[0]]
    return (adjust(R), t), fails


def two_view_init(f1, f2, K):
    F = fundamental(f1, f2)
    pose, fails = decompose(F, K, f1, f2)
    if pose is None:
        return dict(F=F, R=None, t=None, pts=None, fails=fails)
    R, t = pose
    pts = np.array([triangulate(a, b, np.eye(3), np.zeros(3), R, t, K) for a, b in zip(f1, f2)])
    return dict(F=F, R=R, t=t, pts=pts, fails=fails)
