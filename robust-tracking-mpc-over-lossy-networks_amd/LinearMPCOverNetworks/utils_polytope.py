"""Host-side set algebra for the offline stage of the tube-tracking MPC.

Restates, on top of `polytope_lite` + scipy only, the set operations the
reference performs once per model before the first QP is solved
(reference `utils_polytope.py`):

    support                       utils_polytope.py:12-23
    pont_diff                     utils_polytope.py:25-38
    mink_sum / convex hull        utils_polytope.py:40-113, 160-178
    scale                         utils_polytope.py:115-158
    Rakovic eps-mRPI              utils_polytope.py:180-245
    Gilbert-Tan max. output-admissible set   utils_polytope.py:247-268
    Darup-Teichrib RPI            utils_polytope.py:270-414

The outputs (Z, Xc, Uc, Xf) are the (H, h) blocks the HIP solve kernels
consume.  Two things are done differently from the reference, neither of which
changes the resulting sets:

* supports over axis-aligned boxes are evaluated in closed form instead of by
  an LP (the cartpole Darup recursion needs several thousand of them);
* the linear image `-K Z` needed by the tightening step
  (TubeTrackingMPC.py:98-101) is never vertex-enumerated; its support is taken
  as h_{MZ}(a) = h_Z(M^T a).
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import ConvexHull, HalfspaceIntersection

from .polytope_lite import (ABS_TOL, Polytope, as_polytope, box_bounds, lp_max_batch,
                            reduce)


class LinearImage:
    """The set {M x | x in P}, known only through its support function."""

    def __init__(self, P: Polytope, M: np.ndarray):
        self.P = as_polytope(P)
        self.M = np.atleast_2d(np.asarray(M, dtype=np.float64))

    @property
    def dim(self) -> int:
        return self.M.shape[0]


def support_batch(poly, X) -> np.ndarray:
    """h_P(x) = max_{y in P} x^T y for every row x of X: closed form over a box, otherwise one batch of LPs
    (utils_polytope.py:12-23 evaluates them one linprog call at a time)."""
    X = np.atleast_2d(np.asarray(X, dtype=np.float64))
    if isinstance(poly, LinearImage):
        return support_batch(poly.P, X @ poly.M)
    poly = as_polytope(poly)
    bb = getattr(poly, "_box", False)
    if bb is False:
        bb = box_bounds(poly)
        poly._box = bb
    if bb is not None:
        lo, hi = bb
        return np.sum(np.where(X >= 0, X * hi, X * lo), axis=1)
    val, st = lp_max_batch(X, poly.A, poly.b)
    for s1 in st[st != 0]:
        print(f"Status of the support linear program: {s1}")
    return val


def support(poly, x) -> float:
    """h_P(x) = max_{y in P} x^T y  (utils_polytope.py:12-23)."""
    return float(support_batch(poly, np.asarray(x, dtype=np.float64).reshape(1, -1))[0])


def pont_diff(poly1, poly2) -> Polytope:
    """P1 (-) P2 = {x | x + y in P1 for all y in P2}: same rows as P1, offsets
    reduced by the support of P2 along each row (utils_polytope.py:25-38)."""
    poly1 = as_polytope(poly1)
    return Polytope(poly1.A.copy(), poly1.b - support_batch(poly2, poly1.A))


def extreme(poly) -> np.ndarray:
    """Vertices (rows) of a bounded full-dimensional polytope."""
    poly = as_polytope(poly)
    if poly.vertices is not None:
        return poly.vertices.copy()
    n = poly.dim
    if n == 1:
        hi = min(bi / a[0] for a, bi in zip(poly.A, poly.b) if a[0] > 0)
        lo = max(bi / a[0] for a, bi in zip(poly.A, poly.b) if a[0] < 0)
        return np.array([[lo], [hi]])
    # Chebyshev centre as the strictly interior point qhull needs
    nrm = np.linalg.norm(poly.A, axis=1)
    cheb = np.zeros((1, n + 1))
    cheb[0, -1] = 1.0                                   # max r  s.t.  A x + r |a_i| <= b,  r >= 0
    val, st, xc = lp_max_batch(cheb, np.r_[np.c_[poly.A, nrm], np.c_[np.zeros((1, n)), -1.0]], np.r_[poly.b, 0.0], want_x=True)
    if st[0] != 0 or not val[0] > 0:
        raise ValueError("polytope is empty or not full dimensional")
    hs = HalfspaceIntersection(np.c_[poly.A, -poly.b], xc[0, :n])
    V = hs.intersections
    V = V[ConvexHull(V).vertices] if V.shape[0] > n + 1 else V
    poly.vertices = V
    return V.copy()


def determine_convex_hull(vertices: np.ndarray) -> Polytope:
    """H-representation of conv(vertices), one vertex per row (utils_polytope.py:160-178)."""
    vertices = np.asarray(vertices, dtype=np.float64)
    if vertices.shape[1] == 1:
        lo, hi = vertices.min(), vertices.max()
        return Polytope([[1.0], [-1.0]], [hi, -lo], vertices=np.array([[lo], [hi]]))
    hull = ConvexHull(vertices)
    eq = hull.equations
    # qhull returns one facet per simplex; merge coplanar duplicates
    _, idx = np.unique(np.round(eq, 10), axis=0, return_index=True)
    eq = eq[np.sort(idx)]
    return Polytope(eq[:, :-1], -eq[:, -1], vertices=vertices[hull.vertices])


def mink_sum(poly1, poly2) -> Polytope:
    """P1 (+) P2 for a polytope, a point or a vertex list (utils_polytope.py:40-113)."""
    poly1 = as_polytope(poly1)
    if isinstance(poly2, np.ndarray) and poly2.ndim == 1:
        return Polytope(poly1.A, poly1.b + poly1.A @ poly2)
    V1 = extreme(poly1)
    V2 = poly2 if isinstance(poly2, np.ndarray) else extreme(poly2)
    sums = (V1[:, None, :] + V2[None, :, :]).reshape(-1, V1.shape[1])
    return determine_convex_hull(sums)


def scale(poly, scaling_variable):
    """s*P for a scalar, M*P for a matrix (utils_polytope.py:115-158).

    A matrix image with a one-dimensional range is returned as an interval;
    otherwise it is returned as a `LinearImage`, which `support`/`pont_diff`
    accept wherever the reference passes the vertex-enumerated polytope."""
    poly = as_polytope(poly)
    sv = np.asarray(scaling_variable, dtype=np.float64)
    if np.squeeze(sv).ndim == 0:   # the reference treats a 1x1 array as a scalar too
        s = float(np.squeeze(sv))
        if s == 1:
            return poly.copy()
        if s == 0:
            n = poly.dim
            return Polytope(np.r_[np.eye(n), -np.eye(n)], np.zeros(2 * n))
        if s > 0:
            return Polytope(poly.A, s * poly.b,
                            None if poly.vertices is None else s * poly.vertices)
        return Polytope(poly.A / s, poly.b)
    M = np.atleast_2d(sv)
    if M.shape[0] == 1:
        hi = support(poly, M[0])
        lo = -support(poly, -M[0])
        return Polytope([[1.0], [-1.0]], [hi, -lo])
    if poly.vertices is not None or poly.dim <= 3:
        return determine_convex_hull(extreme(poly) @ M.T)
    return LinearImage(poly, M)


def calculate_minimal_robust_positively_invariant_set(A, W, eps_var: float = 1.9e-5, s_max: int = 20):
    """Rakovic et al. eps-outer approximation of the mRPI set of x+ = A x + w
    (utils_polytope.py:180-245).  Returns (F_s / (1 - alpha), status)."""
    A = np.asarray(A, dtype=np.float64)
    W = as_polytope(W)
    if A.shape[0] != A.shape[1]:
        print("A needs to be a square matrix. Returning None")
        return None
    if np.any(W.b <= 0):
        print("The polytope W does not contain the origin. Therefore, we return None")
        return None
    F, g = W.A, W.b
    nx = A.shape[0]
    A_pwr = [np.linalg.matrix_power(A, i) for i in range(s_max)]
    M_pos = np.zeros(nx)
    M_neg = np.zeros(nx)
    status, s, alpha = -1, 0, None
    while s < s_max - 1:
        s += 1
        alpha = max(support(W, A_pwr[s].T @ F[i]) / g[i] for i in range(F.shape[0]))
        for j in range(nx):
            M_pos[j] += support(W, A_pwr[s - 1][j])
            M_neg[j] += support(W, -A_pwr[s - 1][j])
        M_s = max(M_pos.max(), M_neg.max())
        if alpha <= eps_var / (eps_var + M_s):
            status = 0
            break
    if status != 0:
        print(f"In the RPI calculation, we reached the iteration maximum {s_max} without converging!")
        return None, status
    VW = extreme(W)
    Fs = Polytope(W.A, W.b, vertices=VW)
    for i in range(1, s):
        Fs = mink_sum(Fs, VW @ A_pwr[i].T)
    return scale(Fs, 1.0 / (1.0 - alpha)), status


def calculate_maximum_admissible_output_set(A, X, abs_tol: float = ABS_TOL, t_max: int = 100000,
                                            verbose: bool = True) -> Polytope:
    """Gilbert-Tan Algorithm 3.1: O_inf of x+ = A x subject to x(k) in X for all k
    (utils_polytope.py:247-268).

    The reference builds O_{t+1} = O_t /\\ {G A^{t+1} x <= f} and stops when
    O_{t+1} == O_t.  Equality holds exactly when every new row is redundant for
    O_t, which is what is tested here (one LP per new row); rows are appended
    only when they cut, and the final set is reduced once."""
    A = np.asarray(A, dtype=np.float64)
    X = as_polytope(X)
    G, f = X.A, X.b
    H, h = G.copy(), f.copy()
    At = np.eye(A.shape[0])
    t = 0
    while t < t_max:
        At = At @ A
        Gn = G @ At
        live = np.linalg.norm(Gn, axis=1) >= 1e-14
        val, st = lp_max_batch(Gn[live], H, h)         # the rows of one step are tested against the same O_t
        cut = (st != 0) | (val > f[live] + abs_tol)
        newH, newh = Gn[live][cut], f[live][cut]
        added = bool(cut.any())
        if not added:
            if verbose:
                print(f"Admissible set calculation has converged at t = {t}")
            break
        H = np.r_[H, newH]
        h = np.r_[h, newh]
        t += 1
    return reduce(Polytope(H, h), abs_tol)


def calculate_RPI(A, W, X, U, K, eps_var: float = 1e-4, s_max: int = 20,
                  return_container: bool = False, verbose: bool = True):
    """Darup-Teichrib RPI approximation (utils_polytope.py:270-414).

    k* is the first k with (9a) (1+eps) h_W((Hw A^k)_i) <= eps hw_i for all i and
    (9b) (1+eps) sum_{j<k} h_W((Hd A^j)_l) <= hd_l for all l, Hd = [Hx; -Hu K];
    container C = {Hd x <= (1+eps) bc_k}; RPI = {Hd A^i x <= hc - bc_i, i < k*}."""
    A = np.asarray(A, dtype=np.float64)
    W, X, U = as_polytope(W), as_polytope(X), as_polytope(U)
    K = np.atleast_2d(np.asarray(K, dtype=np.float64))
    if A.shape[0] != A.shape[1]:
        print("A needs to be a square matrix. Returning None")
        return None
    if np.any(W.b <= 0):
        print("The polytope W does not contain the origin. Therefore, we return None")
        return None
    status = -1
    Hw, hw = W.A, W.b
    Hd = np.r_[X.A, -U.A @ K]
    hd = np.r_[X.b, U.b]
    nd = Hd.shape[0]
    A_pwr = [np.eye(A.shape[0])]
    for _ in range(s_max):
        A_pwr.append(A_pwr[-1] @ A)
    bc_all = np.zeros((nd, s_max))
    k_star, found = 1, False
    while k_star < s_max and not found:
        HwAk = Hw @ A_pwr[k_star]
        HdAj = Hd @ A_pwr[k_star - 1]
        cond_a = bool(np.all((1 + eps_var) * support_batch(W, HwAk) <= eps_var * hw))
        inc = support_batch(W, HdAj)
        bc_all[:, k_star - 1] = inc if k_star == 1 else bc_all[:, k_star - 2] + inc
        cond_b = bool(np.all((1 + eps_var) * bc_all[:, k_star - 1] <= hd))
        if not cond_b:
            # the partial sums only grow with k (0 is in W): once (9b) fails it fails for every larger k and every larger
            # s_max.  The reference keeps multiplying s_max by ten here and never returns (TubeRegulatorMPC.py:48-71).
            print("The disturbance set is too large for the constraints: no RPI set fits inside them (status -2).")
            return None, -2
        if cond_a and cond_b:
            found, status = True, 0
        else:
            k_star += 1
    if verbose:
        print(f"k_star = {k_star}")
    if not found:
        print(f"In the RPI calculation, we reached the iteration maximum {s_max} without converging!")
        return None, status
    hc = (1 + eps_var) * bc_all[:, k_star - 1]
    C = Polytope(Hd, hc)
    HcAk = Hd @ A_pwr[k_star]
    if not np.all((1 + eps_var) * support_batch(C, HcAk) <= eps_var * hc):
        # does not depend on s_max: status -3 lets the caller stop instead of retrying with a larger s_max for ever
        # (the reference returns -1 here and its retry loop never ends, TubeRegulatorMPC.py:48-71)
        print("The container set C does not fulfill the condition for calculating the RPI. Returning None")
        return None, -3
    H_rows = [Hd]
    h_rows = [hc]
    for i in range(1, k_star):
        H_rows.append(Hd @ A_pwr[i])
        h_rows.append(hc - bc_all[:, i - 1])
    rpi = Polytope(np.vstack(H_rows), np.concatenate(h_rows))
    rpi.k_star = k_star
    if return_container:
        return rpi, C, status
    return rpi, status


def project_polytope(P, E, tol: float = 1e-9, max_vertices: int = 2000) -> Polytope:
    """Exact H-representation of the linear image {E x : x in P} of a bounded polytope for a
    low-dimensional image space (k = E.shape[0] <= 4): convex-hull method.  Support points of the
    image are found by LPs over P; the hull of the points found so far is refined until every one
    of its facets is a supporting hyperplane of the image (checked by one LP per facet).

    Not in the reference (it keeps the projected-out variables in the QP, TubeTrackingMPC.py:293);
    used to eliminate the free auxiliaries of the packet-received problem at set-up time."""
    P = as_polytope(P)
    E = np.atleast_2d(np.asarray(E, dtype=np.float64))
    k = E.shape[0]

    def sup_batch(directions):
        directions = np.atleast_2d(directions)
        _, st, xs = lp_max_batch(directions @ E, P.A, P.b, want_x=True)
        if np.any(st != 0):
            raise ValueError("projection: LP failed (status %d); polytope unbounded or empty?" % st[st != 0][0])
        return xs @ E.T

    def sup(direction):
        return sup_batch(direction)[0]

    if k == 1:
        hi, lo = sup(np.array([1.0]))[0], sup(np.array([-1.0]))[0]
        return Polytope([[1.0], [-1.0]], [hi, -lo], vertices=np.array([[lo], [hi]]))
    pts = list(sup_batch(np.r_[np.eye(k), -np.eye(k)]))
    rng = np.random.default_rng(0)
    while np.linalg.matrix_rank(np.array(pts)[1:] - pts[0], tol=1e-9) < k:
        if len(pts) > 2 * k + 50:
            raise ValueError("projection is not full dimensional")
        d = rng.standard_normal(k)
        pts.append(sup(d / np.linalg.norm(d)))
    pts = np.unique(np.round(np.array(pts), 12), axis=0)
    verified = set()
    while True:
        hull = ConvexHull(pts)
        eq = hull.equations
        _, idx = np.unique(np.round(eq, 10), axis=0, return_index=True)
        eq = eq[np.sort(idx)]
        added = False
        todo = [row for row in eq if tuple(np.round(row, 9)) not in verified]
        ys = sup_batch(np.array([row[:-1] for row in todo])) if todo else []
        for row, y in zip(todo, ys):
            a, b = row[:-1], -row[-1]
            if a @ y > b + tol * max(1.0, abs(b)):
                pts = np.vstack([pts, y])
                added = True
            else:
                verified.add(tuple(np.round(row, 9)))
        if not added:
            return Polytope(eq[:, :-1], -eq[:, -1], vertices=pts[hull.vertices])
        if len(pts) > max_vertices:
            raise ValueError("projection: more than %d vertices" % max_vertices)


def steady_state_basis(A, B) -> np.ndarray:
    """Orthonormal basis N ((nx+nu) x nu) of {(x_bar, u_bar) : (A - I) x_bar + B u_bar = 0}
    (TubeTrackingMPC.py:147)."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64).reshape(A.shape[0], -1)
    M = np.c_[A - np.eye(A.shape[0]), B]
    _, s, Vt = np.linalg.svd(M)
    rank = int(np.sum(s > 1e-10 * s[0]))
    return Vt[rank:].T


def eliminate_terminal_auxiliaries(Xf, A, B) -> Polytope:
    """Terminal constraint of the packet-received problem with its free variables eliminated.

    TubeTrackingMPC.py:293 writes  HT_x x_N' + HT_xbar x_bar + HT_u u_bar' <= hT  on the BASE problem's
    x_N' and u_bar', which the packet-received problem neither constrains otherwise nor prices: they are
    free auxiliaries and the row block only says  x_bar in proj_xbar(Xf).  Together with the steady-state
    equation (x_bar, u_bar) = N phi this is a polytope in phi (dimension nu), computed here exactly and
    returned as rows acting on [x_bar; u_bar] (valid on the steady-state subspace)."""
    Xf = as_polytope(Xf)
    A = np.asarray(A, dtype=np.float64)
    nx = A.shape[0]
    Nss = steady_state_basis(A, B)
    nu = Nss.shape[0] - nx
    HT = Xf.A
    # variables (x_aux, u_aux, phi)
    rows = np.c_[HT[:, :nx], HT[:, 2 * nx:], HT[:, nx:2 * nx] @ Nss[:nx]]
    E = np.c_[np.zeros((Nss.shape[1], nx + nu)), np.eye(Nss.shape[1])]
    Pphi = project_polytope(Polytope(rows, Xf.b), E)
    return Polytope(Pphi.A @ Nss.T, Pphi.b)
