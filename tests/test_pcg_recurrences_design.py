"""The arithmetic of pg_pcg_persistent_kernel (pg_engine.hip) restated in numpy, without a GPU: the Chronopoulos-Gear form of
preconditioned CG -- w = A u with u = M^-1 r, p and s = A p by recurrence, both dot products of a step in ONE reduction -- with the
coarse residual of the two-level preconditioner carried by RECURRENCE (r_c <- r_c - alpha P^T s, P^T s = P^T w + beta P^T s), so that
an iteration needs two exchanges (the neighbours' u; nine numbers per group) and no gather for the restriction.  Checked here:
the iterates are those of textbook PCG with the same preconditioner, the carried coarse residual stays P^T r, and the stopping
test on the recurrence residual stops at the same iteration."""
import numpy as np


def _system(rng, nodes=48, group=8):
    n = 6 * nodes
    # block-sparse SPD matrix: a chain plus a few long edges (a pose graph's J^T J + D)
    A = np.zeros((n, n))
    edges = [(i, i + 1) for i in range(nodes - 1)] + [(int(a), int(b)) for a, b in rng.integers(0, nodes, (nodes, 2)) if a != b]
    for i, j in edges:
        Ji, Jj = rng.normal(size=(6, 6)), rng.normal(size=(6, 6))
        J = np.zeros((6, n)); J[:, 6 * i:6 * i + 6] = Ji; J[:, 6 * j:6 * j + 6] = Jj
        A += J.T @ J
    A += np.diag(rng.uniform(0.01, 0.1, n))
    Minv = np.zeros((n, n))
    for i in range(nodes):
        sl = slice(6 * i, 6 * i + 6)
        Minv[sl, sl] = np.linalg.inv(A[sl, sl])
    na = nodes // group
    P = np.zeros((n, 6 * na))
    for i in range(nodes):
        P[6 * i:6 * i + 6, 6 * (i // group):6 * (i // group) + 6] = np.eye(6) + 0.1 * rng.normal(size=(6, 6))
    Ainv = np.linalg.inv(P.T @ A @ P)
    return A, Minv, P, Ainv, rng.normal(size=n)


def _pcg(A, prec, b, tol, iters):
    x = np.zeros_like(b); r = b.copy(); z = prec(r); p = z.copy(); rz = r @ z
    xs = []
    for k in range(iters):
        q = A @ p
        alpha = rz / (p @ q)
        x = x + alpha * p; r = r - alpha * q
        xs.append(x.copy())
        if r @ r <= tol * tol * (b @ b):
            break
        z = prec(r); rz_new = r @ z
        p = z + (rz_new / rz) * p; rz = rz_new
    return xs


def _cg_cg(A, Minv, P, Ainv, b, tol, iters):
    """what every group computes: r_c, P^T s are the replicated 6 n_groups vectors; (gamma, delta, rr, P^T w) is exchange B"""
    x = np.zeros_like(b); r = b.copy()
    rc = P.T @ r                       # (the start: one gather of P^T r)
    pts = np.zeros_like(rc)
    u = Minv @ r + P @ (Ainv @ rc)
    p = np.zeros_like(b); s = np.zeros_like(b)
    g_old = a_old = 1.0
    xs, drift = [], []
    for k in range(iters + 1):
        w = A @ u                      # exchange A: the neighbours' u
        gamma, delta, rr, ptw = r @ u, w @ u, r @ r, P.T @ w          # exchange B: ONE reduction + the groups' entries of P^T w
        if k > 0 and rr <= tol * tol * (b @ b):
            break
        if k >= iters:
            break
        beta = 0.0 if k == 0 else gamma / g_old
        alpha = gamma / (delta if k == 0 else delta - beta * gamma / a_old)
        p = u + beta * p; s = w + beta * s
        x = x + alpha * p; r = r - alpha * s
        pts = ptw + beta * pts; rc = rc - alpha * pts                 # the coarse residual by recurrence: no gather
        drift.append(np.abs(rc - P.T @ r).max() / np.abs(P.T @ b).max())      # (against the start: rounding does not shrink with the residual)
        u = Minv @ r + P @ (Ainv @ rc)
        g_old, a_old = gamma, alpha
        xs.append(x.copy())
    return xs, drift


def test_two_exchange_pcg_is_textbook_pcg():
    rng = np.random.default_rng(7)
    for tol in (1e-3, 1e-10):
        A, Minv, P, Ainv, b = _system(rng)
        prec = lambda v: Minv @ v + P @ (Ainv @ (P.T @ v))          # noqa: E731
        ref = _pcg(A, prec, b, tol, 400)
        got, drift = _cg_cg(A, Minv, P, Ainv, b, tol, 400)
        assert len(got) == len(ref) and len(ref) < 400               # the same iteration stops both
        sol = np.linalg.solve(A, b)
        for xa, xb in zip(got, ref):
            assert np.abs(xa - xb).max() <= 1e-8 * np.abs(sol).max()
        assert max(drift) < 1e-12                                    # the carried coarse residual IS P^T r, to rounding
        assert np.abs(A @ got[-1] - b).max() <= 20 * tol * np.abs(b).max()
