"""Closed-form Zhang initialisation (numpy) used to feed the calibration Gauss-Newton with the
same start the reference uses: DLT homographies -> intrinsics -> extrinsics.
Follows st3-calibration/src/src/calib.cpp:55-93 (computeHomoMat), :95-140
(reconstructIntriMat), :142-173 (reconstructExtriMat); corner files are parsed like
cbcorner.cpp:50-73 (std::stof: values are float-rounded before widening) and read in sorted
order (helper.cpp:9).  Test-side helper for the golden calibration fixture."""
import glob
import os

import numpy as np


def read_corners(directory, square):
    files = sorted(glob.glob(os.path.join(directory, "*.txt")))      # helper.cpp:9 std::sort on paths
    obj, img = [], []
    for f in files:
        with open(f) as fh:
            lines = fh.read().strip().splitlines()
        rows, cols = (int(v) for v in lines[0].split(","))
        im = np.zeros((rows, cols, 2))
        for ln in lines[1:]:
            r, c, x, y = ln.split(",")
            im[int(r), int(c)] = [np.float32(x), np.float32(y)]       # std::stof
        jj, ii = np.meshgrid(np.arange(cols), np.arange(rows))
        ob = np.stack([jj * square, ii * square], -1)                  # calib.cpp:28
        obj.append(ob.reshape(-1, 2)); img.append(im.reshape(-1, 2))
    return np.array(obj), np.array(img)


def homography(img, obj):
    n = len(img)
    A = np.zeros((2 * n, 9))
    x, y, u, v = obj[:, 0], obj[:, 1], img[:, 0], img[:, 1]
    A[0::2, 0] = x; A[0::2, 1] = y; A[0::2, 2] = 1; A[0::2, 6] = -u * x; A[0::2, 7] = -u * y; A[0::2, 8] = -u
    A[1::2, 3] = x; A[1::2, 4] = y; A[1::2, 5] = 1; A[1::2, 6] = -v * x; A[1::2, 7] = -v * y; A[1::2, 8] = -v
    _, _, Vt = np.linalg.svd(A)
    return Vt[-1].reshape(3, 3)


def intrinsics(Hs):
    def cof(H, i, j):
        hi, hj = H[:, i], H[:, j]
        return np.array([hi[0] * hj[0], hi[2] * hj[0] + hi[0] * hj[2], hi[1] * hj[1],
                         hi[2] * hj[1] + hi[1] * hj[2], hi[2] * hj[2]])
    C = []
    for H in Hs:
        C.append(cof(H, 0, 1))
        C.append(cof(H, 0, 0) - cof(H, 1, 1))
    _, _, Vt = np.linalg.svd(np.array(C))
    b11, b13, b22, b23, b33 = Vt[-1]
    v0 = -b23 / b22
    lam = b33 - (b13 * b13 - v0 * b11 * b23) / b11
    alpha = np.sqrt(lam / b11)
    beta = np.sqrt(lam / b22)
    u0 = -b13 * alpha * alpha / lam
    return alpha, beta, u0, v0


def extrinsics(Hs, K):
    Ki = np.linalg.inv(K)
    out = []
    for H in Hs:
        r1 = Ki @ H[:, 0]; r2 = Ki @ H[:, 1]
        lam = 1.0 / (2.0 * np.linalg.norm(r1)) + 1.0 / (2.0 * np.linalg.norm(r2))
        r1 = r1 / np.linalg.norm(r1); r2 = r2 / np.linalg.norm(r2)
        r3 = np.cross(r1, r2)
        r1 = np.cross(r2, r3)
        t = lam * Ki @ H[:, 2]
        R = np.stack([r1, r2, r3], 1)
        U, _, Vt = np.linalg.svd(R)
        R = U @ Vt
        if t[2] < 0:            # SVD sign ambiguity of the homography: keep the board in front
            R = np.stack([-R[:, 0], -R[:, 1], R[:, 2]], 1); t = -t
        out.append((R, t))
    return out


def zhang_init(obj, img, se3_log):
    Hs = [homography(img[v], obj[v]) for v in range(len(obj))]
    alpha, beta, u0, v0 = intrinsics(Hs)
    K = np.array([[alpha, 0, u0], [0, beta, v0], [0, 0, 1.0]])
    ext = extrinsics(Hs, K)
    params = np.zeros(9 + 6 * len(obj))
    params[:4] = [alpha, beta, u0, v0]
    for v, (R, t) in enumerate(ext):
        params[9 + 6 * v: 15 + 6 * v] = se3_log(R, t)
    return params
