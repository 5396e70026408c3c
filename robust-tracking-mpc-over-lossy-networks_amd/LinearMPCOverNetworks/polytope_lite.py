"""Minimal H-representation polytope used by the host-side set-up stage.

The reference hands `polytope.Polytope` objects to its controllers
(`RegulatorMPC.py:33-43`) and reads `.A`, `.b` back from the computed sets
(`results_linear_system.py:123`, `TubeTrackingMPC.py:110-113`).  The `polytope`
package is not part of this image, so the host code here accepts any object
exposing `.A` / `.b` (or an `(A, b)` tuple) and returns this light class, which
keeps the three behaviours the reference relies on:

* `x in P`         -> `A x - b <= abs_tol` with abs_tol 1e-7 (results_linear_system.py:258)
* `P.A`, `P.b`     -> float64 arrays, `b` flat
* `reduce(P)`      -> LP-based removal of redundant rows (TubeRegulatorMPC.py:74)

Everything here runs once per model on the host; nothing in this file is on
the per-timestep hot path.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import linprog

ABS_TOL = 1e-7


class Polytope:
    """{x | A x <= b}."""

    def __init__(self, A, b, vertices=None, normalize=False):
        A = np.array(A, dtype=np.float64)
        if A.ndim == 1:
            A = A.reshape(1, -1)
        b = np.array(b, dtype=np.float64).reshape(-1)
        if A.shape[0] != b.shape[0]:
            raise ValueError(f"A has {A.shape[0]} rows but b has {b.shape[0]} entries")
        if normalize:
            nrm = np.linalg.norm(A, axis=1)
            keep = nrm > 1e-14
            A, b, nrm = A[keep], b[keep], nrm[keep]
            A = A / nrm[:, None]
            b = b / nrm
        self.A = A
        self.b = b
        self.vertices = None if vertices is None else np.array(vertices, dtype=np.float64)

    @property
    def dim(self) -> int:
        return self.A.shape[1]

    def copy(self) -> "Polytope":
        return Polytope(self.A.copy(), self.b.copy(),
                        None if self.vertices is None else self.vertices.copy())

    def contains(self, x, abs_tol: float = ABS_TOL):
        """Row-wise membership test; `x` is (dim,), (dim,1) or (dim,K)."""
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == 1:
            return bool(np.all(self.A @ x - self.b <= abs_tol))
        if x.shape[0] != self.dim:
            raise ValueError("points must be stacked column-wise")
        res = np.all(self.A @ x - self.b[:, None] <= abs_tol, axis=0)
        return bool(res[0]) if x.shape[1] == 1 else res

    def __contains__(self, x) -> bool:
        r = self.contains(x)
        return bool(np.all(r))

    def __repr__(self) -> str:
        return f"Polytope(rows={self.A.shape[0]}, dim={self.dim})"


def as_polytope(P) -> Polytope:
    """Accept a Polytope, anything with .A/.b, or an (A, b) pair."""
    if isinstance(P, Polytope):
        return P
    if hasattr(P, "A") and hasattr(P, "b"):
        return Polytope(P.A, P.b, getattr(P, "vertices", None))
    A, b = P
    return Polytope(A, b)


def box2poly(bounds) -> Polytope:
    """[[lo, hi], ...] -> box polytope, rows ordered [+I; -I]."""
    bounds = np.asarray(bounds, dtype=np.float64)
    n = bounds.shape[0]
    A = np.r_[np.eye(n), -np.eye(n)]
    b = np.r_[bounds[:, 1], -bounds[:, 0]]
    return Polytope(A, b)


def box_bounds(P: Polytope):
    """If every row of P is +-e_i (times a positive scale) return (lo, hi), else None."""
    n = P.dim
    lo = np.full(n, -np.inf)
    hi = np.full(n, np.inf)
    for a, bi in zip(P.A, P.b):
        nz = np.flatnonzero(a)
        if nz.size != 1:
            return None
        j = nz[0]
        if a[j] > 0:
            hi[j] = min(hi[j], bi / a[j])
        else:
            lo[j] = max(lo[j], bi / a[j])
    if not (np.all(np.isfinite(lo)) and np.all(np.isfinite(hi))):
        return None
    return lo, hi


def _lp_max(c, A, b):
    """max c^T x s.t. A x <= b (free x). Returns (value, status)."""
    res = linprog(-np.asarray(c, dtype=np.float64).reshape(-1), A_ub=A, b_ub=b,
                  bounds=(None, None), method="highs")
    if res.status != 0:
        return np.inf if res.status == 3 else np.nan, res.status
    return -res.fun, 0


def reduce(P: Polytope, abs_tol: float = ABS_TOL) -> Polytope:
    """Remove redundant rows: row i is dropped when max a_i x over the other rows
    (with b_i relaxed by one) does not exceed b_i + abs_tol."""
    P = Polytope(P.A, P.b, normalize=True)
    A, b = P.A, P.b
    keep = np.ones(len(b), dtype=bool)
    for i in range(len(b)):
        keep[i] = False
        if not keep.any():
            keep[i] = True
            continue
        Ai = np.r_[A[keep], A[i:i + 1]]
        bi = np.r_[b[keep], b[i] + 1.0]
        val, st = _lp_max(A[i], Ai, bi)
        if st != 0 or val > b[i] + abs_tol:
            keep[i] = True
    return Polytope(A[keep], b[keep])


def is_subset(P: Polytope, Q: Polytope, abs_tol: float = ABS_TOL) -> bool:
    """P subset of Q  <=>  support_P(q_i) <= b_i for every row of Q."""
    for a, bi in zip(Q.A, Q.b):
        val, st = _lp_max(a, P.A, P.b)
        if st != 0 or val > bi + abs_tol:
            return False
    return True
