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

Everything here runs once per model; nothing in this file is on the per-timestep
hot path.  The linear programs behind the set operations are all support
functions of one polytope along many directions; they go through `lp_max_batch`,
which has two back-ends chosen with `set_lp_backend`:

* "hip"   (default) one launch of the batched LP kernel per set operation
          (include/tmpc.h: tmpc_lp_batch, csrc/tmpc_lp.hip), dimension <= 32;
* "scipy" one `scipy.optimize.linprog(method="highs")` call per LP -- what the
          reference does (utils_polytope.py:19); the oracle of the LP kernel in
          the tests, and the back-end for hosts without a GPU.

There is no silent switch between the two: a missing GPU or library raises.
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


_LP_BACKEND = "hip"
_LP_DEVICE = 0
LP_MAX_DIM_HIP = 32


def set_lp_backend(name: str, device: int = 0) -> str:
    """Choose who solves the set-up LPs: "hip" (batched kernel) or "scipy" (HiGHS, one call per LP).
    Returns the previous choice."""
    global _LP_BACKEND, _LP_DEVICE
    if name not in ("hip", "scipy"):
        raise ValueError('lp backend must be "hip" or "scipy"')
    old = _LP_BACKEND
    _LP_BACKEND, _LP_DEVICE = name, int(device)
    return old


def get_lp_backend() -> str:
    return _LP_BACKEND


def _lp_max(c, A, b):
    """max c^T x s.t. A x <= b (free x) by HiGHS (utils_polytope.py:19). Returns (value, status)."""
    res = linprog(-np.asarray(c, dtype=np.float64).reshape(-1), A_ub=A, b_ub=b,
                  bounds=(None, None), method="highs")
    if res.status != 0:
        return np.inf if res.status == 3 else np.nan, res.status
    return -res.fun, 0


def lp_max_batch(C, A, b, relax=None, relax_by: float = 1.0, want_x: bool = False):
    """val[k] = max C[k] . x  s.t.  A x <= b, with row relax[k] of b raised by relax_by (relax[k] = -1: none).

    Returns (val, status[, x]); status uses scipy's codes: 0 solved, 2 infeasible, 3 unbounded, 4 numerical
    trouble.  val is +inf when unbounded and nan for the other failures."""
    C = np.atleast_2d(np.asarray(C, dtype=np.float64))
    A = np.asarray(A, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    nb = C.shape[0]
    rel = np.full(nb, -1, dtype=np.int32) if relax is None else np.asarray(relax, dtype=np.int32).reshape(-1)
    if _LP_BACKEND == "hip":
        if A.shape[1] > LP_MAX_DIM_HIP:
            raise ValueError(f"the batched LP kernel covers dimension <= {LP_MAX_DIM_HIP}; this polytope has "
                             f'dimension {A.shape[1]}: call set_lp_backend("scipy") for it')
        from . import _native
        out = _native.lp_batch(A, b, C, relax=rel, relax_by=relax_by, device=_LP_DEVICE, want_x=want_x)
        st = np.select([out["status"] == 0, out["status"] == 1, out["status"] == 2, out["status"] == 4], [0, 1, 2, 3], default=4).astype(int)
        # kernel status 1: the interior-point iterate was returned without its certificate (feasible to 1e-11, multipliers
        # >= 0, dual residual <= 1e-11).  Round 3: none of the 7 458 LPs of a cartpole model, 1 % of the 2 092 of the
        # synthetic one (d = 28: c keeps ~1e-8 along a long optimal face, below what double precision resolves there);
        # the value is within ~1e-8.  These values feed the constraint tightening, so such an LP is solved again by HiGHS
        # -- the call the reference makes for every LP (utils_polytope.py:19) -- instead of being passed on as "solved".
        for k in np.flatnonzero(st == 1):
            bk = b
            if rel[k] >= 0:
                bk = b.copy()
                bk[rel[k]] += relax_by
            res = linprog(-C[k], A_ub=A, b_ub=bk, bounds=(None, None), method="highs")
            st[k] = res.status
            out["val"][k] = -res.fun if res.status == 0 else (np.inf if res.status == 3 else np.nan)
            if want_x:
                out["x"][k] = res.x if res.status == 0 else np.nan
        return (out["val"], st, out["x"]) if want_x else (out["val"], st)
    val = np.empty(nb)
    st = np.zeros(nb, dtype=int)
    xs = np.full((nb, A.shape[1]), np.nan)
    for k in range(nb):
        bk = b
        if rel[k] >= 0:
            bk = b.copy()
            bk[rel[k]] += relax_by
        res = linprog(-C[k], A_ub=A, b_ub=bk, bounds=(None, None), method="highs")
        st[k] = res.status
        if res.status == 0:
            val[k] = -res.fun
            xs[k] = res.x
        else:
            val[k] = np.inf if res.status == 3 else np.nan
    return (val, st, xs) if want_x else (val, st)


def reduce(P: Polytope, abs_tol: float = ABS_TOL) -> Polytope:
    """Remove redundant rows: row i is dropped when max a_i x over the other rows
    (with b_i relaxed by one) does not exceed b_i + abs_tol.

    The reference's `polytope.reduce` walks the rows one LP at a time.  Here ONE batch tests every row against
    all the others: a row that cuts even then cuts for any subset (kept), a row that is slack by more than
    abs_tol is never tight on the set, and all such rows can leave together without changing the set.  Only the
    rows within abs_tol of their bound (ties, near-duplicates) are order dependent; they are walked in index
    order like the reference does."""
    P = Polytope(P.A, P.b, normalize=True)
    A, b = P.A, P.b
    n = len(b)
    if n <= 1:
        return P
    val, st = lp_max_batch(A, A, b, relax=np.arange(n), relax_by=1.0)
    cuts = (st != 0) | (val > b + abs_tol)
    slack = (st == 0) & (val < b - abs_tol)
    keep = ~slack
    for i in np.flatnonzero(~cuts & ~slack):
        keep[i] = False
        if not keep.any():
            keep[i] = True
            continue
        Ai = np.r_[A[keep], A[i:i + 1]]
        bi = np.r_[b[keep], b[i] + 1.0]
        v, s1 = lp_max_batch(A[i], Ai, bi)
        if s1[0] != 0 or v[0] > b[i] + abs_tol:
            keep[i] = True
    return Polytope(A[keep], b[keep])


def is_subset(P: Polytope, Q: Polytope, abs_tol: float = ABS_TOL) -> bool:
    """P subset of Q  <=>  support_P(q_i) <= b_i for every row of Q."""
    val, st = lp_max_batch(Q.A, P.A, P.b)
    return bool(np.all((st == 0) & (val <= Q.b + abs_tol)))
