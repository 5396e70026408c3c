"""ORACLE (test infrastructure, not product code): dense primal-dual
interior-point solve of   min 1/2 v'Pv + q'v  s.t.  A v = b,  G v <= h   in numpy.

The reference delegates this step to Clarabel through cvxpy
(`self._prob.solve(solver=CLARABEL, tol_gap_abs=1e-7, tol_gap_rel=1e-7)`,
reference TubeTrackingMPC.py:183): an interior-point method on the un-condensed
problem.  Clarabel (Rust, un-pinned in the reference's setup.cfg:13-20) is not
available here, so this file restates the published method class -- Mehrotra
predictor-corrector on the slack form G v + s = h, s >= 0 -- and runs it to a much
tighter tolerance (1e-10) than the reference's 1e-7, so that any difference to the
HIP kernels is attributable to them.  PARITY UNPINNED (see qp_sparse.py).

oracle/tmpc_oracle.c is the same algorithm in C for batches; this version is the
readable one and the checker's checker.
"""
from __future__ import annotations

import numpy as np


def solve_qp(P, q, A, b, G, h, tol: float = 1e-10, max_iter: int = 100, verbose: bool = False):
    """Returns dict(v, y, lam, s, status, iters).  status 0 optimal, 1 max-iter,
    2 primal infeasible (Farkas certificate), 3 numerical failure."""
    P = np.asarray(P, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64)
    G = np.asarray(G, dtype=np.float64)
    h = np.asarray(h, dtype=np.float64)
    n = q.size
    A = np.zeros((0, n)) if A is None else np.asarray(A, dtype=np.float64)
    b = np.zeros(0) if b is None else np.asarray(b, dtype=np.float64)
    me, mi = A.shape[0], G.shape[0]

    # row equilibration of the inequalities keeps the barrier well scaled
    gn = np.maximum(np.linalg.norm(G, axis=1), 1e-12)
    G = G / gn[:, None]
    h = h / gn

    def kkt_solve(d, r1, r2):
        M = P + G.T @ (d[:, None] * G)
        K = np.block([[M, A.T], [A, np.zeros((me, me))]]) if me else M
        sol = np.linalg.solve(K, np.r_[r1, r2])
        return sol[:n], sol[n:]

    # start: minimise the objective plus 1/2 |Gv-h|^2 subject to the equalities
    v, y = kkt_solve(np.ones(mi), -q + G.T @ h, b)
    s = h - G @ v
    shift = max(0.0, -1.5 * s.min()) if s.min() < 1e-2 else 0.0
    s = s + shift if shift > 0 else np.maximum(s, 1e-2)
    lam = np.ones(mi)
    status = 1
    qn = max(1.0, np.max(np.abs(q)))
    hn = max(1.0, np.max(np.abs(h)))
    it = 0
    for it in range(max_iter):
        r_d = P @ v + q + A.T @ y + G.T @ lam
        r_e = A @ v - b
        r_p = G @ v + s - h
        mu = s @ lam / mi
        res_d = np.max(np.abs(r_d)) / qn
        res_p = max(np.max(np.abs(r_p)), np.max(np.abs(r_e)) if me else 0.0) / hn
        if verbose:
            print(f"{it:3d} mu={mu:.3e} rd={res_d:.3e} rp={res_p:.3e}")
        if res_d <= tol and res_p <= tol and mu <= tol:
            status = 0
            break
        # Farkas: lam >= 0 (plus y) with G'lam + A'y ~ 0 and h'lam + b'y < 0
        ln = np.max(lam)
        if ln > 1e8:
            cert = G.T @ lam + A.T @ y
            if np.max(np.abs(cert)) <= 1e-8 * ln and (h @ lam + b @ y) < -1e-8 * ln:
                status = 2
                break
        d = lam / s
        try:
            dv_a, dy_a = kkt_solve(d, -r_d - G.T @ (d * r_p - lam), -r_e)
        except np.linalg.LinAlgError:
            status = 3
            break
        ds_a = -r_p - G @ dv_a
        dl_a = -lam - d * ds_a
        a_aff = _step(s, ds_a, lam, dl_a, 1.0)
        mu_aff = (s + a_aff * ds_a) @ (lam + a_aff * dl_a) / mi
        sigma = (mu_aff / mu) ** 3
        rc = s * lam + ds_a * dl_a - sigma * mu
        dv, dy = kkt_solve(d, -r_d - G.T @ (d * r_p - rc / s), -r_e)
        ds = -r_p - G @ dv
        dl = -(rc + lam * ds) / s
        a = _step(s, ds, lam, dl, 0.995)
        v = v + a * dv
        y = y + a * dy
        s = s + a * ds
        lam = lam + a * dl
        if not (np.all(np.isfinite(v)) and np.all(np.isfinite(lam))):
            status = 3
            break
    return dict(v=v, y=y, lam=lam / gn, s=s * gn, status=status, iters=it)


def _step(s, ds, lam, dl, tau):
    a = 1.0
    m = ds < 0
    if m.any():
        a = min(a, tau * np.min(-s[m] / ds[m]))
    m = dl < 0
    if m.any():
        a = min(a, tau * np.min(-lam[m] / dl[m]))
    return a
