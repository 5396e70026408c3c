"""ORACLE (test infrastructure, not product code): the tube-tracking QP in the
reference's own, un-condensed variables, as plain numpy.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  PARITY UNPINNED: the reference repository holds no golden vectors or
known-answer tests for the QP solution and its solver stack (cvxpy -> Clarabel)
is not installable here, so this restatement is pinned only by (i) a KKT
certificate of every solution it is used to check and (ii) an independent
scipy cross-check at small sizes (tests/test_oracle.py).

`build_sparse_qp` follows `TubeTrackingMPC.generate_optimization_problem`
(reference src/LinearMPCOverNetworks/TubeTrackingMPC.py:104-156) line by line:

    variables   x_mpc (nx,N+1), u_mpc (nu,N), x_bar (nx), u_bar (nu)      :117-120
    initial     x_init - x_0 == 0            (fixed_initial_state)         :127
                Hz (x_init - x_0) <= hz      (otherwise)                   :132
    stage i<N   cost (x_i-x_bar)'Q(x_i-x_bar) + (u_i-u_bar)'R(u_i-u_bar)   :136
                x_{i+1} == A x_i + B u_i                                    :138
                Hx x_i <= hx ,  Hu u_i <= hu                                :139-140
    terminal    cost (x_N-x_bar)'P(x_N-x_bar) + (x_bar-ref)'T(x_bar-ref)   :143-144
                (A-I) x_bar + B u_bar == 0                                  :147
                HT[:, :nx] x_N + HT[:, nx:2nx] x_bar + HT[:, 2nx:] u_bar <= hT   :149

and, for the packet-received problem of `ExtendedTubeTrackingMPC`
(:253-299), the same with Z (-) W in the initial constraint (:266-278) and the
terminal row written literally as in :293, i.e. on `x_mpc[:, N]` and `u_bar` of
the *base* problem, which are free auxiliary variables of this problem.

The QP is returned in the standard form
        min 1/2 v'Pv + q'v + c0   s.t.  Aeq v = beq,  G v <= h.
"""
from __future__ import annotations

import numpy as np


class Layout:
    """Index bookkeeping for v = [x_0..x_N | u_0..u_{N-1} | x_bar | u_bar | (x_aux | u_aux)]."""

    def __init__(self, nx, nu, N, aux=False):
        self.nx, self.nu, self.N, self.aux = nx, nu, N, aux
        self.ox = 0
        self.ou = nx * (N + 1)
        self.oxb = self.ou + nu * N
        self.oub = self.oxb + nx
        self.oxa = self.oub + nu
        self.oua = self.oxa + nx
        self.nvar = self.oub + nu + ((nx + nu) if aux else 0)

    def x(self, i):
        return slice(self.ox + i * self.nx, self.ox + (i + 1) * self.nx)

    def u(self, i):
        return slice(self.ou + i * self.nu, self.ou + (i + 1) * self.nu)

    @property
    def xbar(self):
        return slice(self.oxb, self.oxb + self.nx)

    @property
    def ubar(self):
        return slice(self.oub, self.oub + self.nu)

    @property
    def xaux(self):
        return slice(self.oxa, self.oxa + self.nx)

    @property
    def uaux(self):
        return slice(self.oua, self.oua + self.nu)


def build_sparse_qp(p: dict, x_k, ref, variant: int = 0):
    """p: the problem dict of TubeTrackingMPC._problem_dict().  variant 1 = the
    packet-received problem (needs p['HZW'], p['hZW'])."""
    nx, nu, N = int(p["nx"]), int(p["nu"]), int(p["N"])
    A, B, Q, R, Pm, T = (np.asarray(p[k], dtype=np.float64) for k in ("A", "B", "Q", "R", "P", "T"))
    Hx, hx, Hu, hu = (np.asarray(p[k], dtype=np.float64) for k in ("Hx", "hx", "Hu", "hu"))
    HT = np.asarray(p["HT"], dtype=np.float64) if p.get("HT") is not None else np.zeros((0, 2 * nx + nu))
    hT = np.asarray(p["hT"], dtype=np.float64) if p.get("hT") is not None else np.zeros(0)
    x_k = np.asarray(x_k, dtype=np.float64).reshape(nx)
    ref = np.asarray(ref, dtype=np.float64).reshape(nx)
    received = variant == 1
    # packet-received problem: the literal row :293 acts on free auxiliaries; with p['HTP'] (their
    # elimination, include/tmpc.h) the rows act on [x_bar; u_bar] instead and no auxiliaries exist
    projected = received and p.get("HTP") is not None and int(p.get("literal_terminal_row", 1)) == 1
    L = Layout(nx, nu, N, aux=received and not projected)
    nv = L.nvar

    def sel(s):
        E = np.zeros((s.stop - s.start, nv))
        E[np.arange(s.stop - s.start), np.arange(s.start, s.stop)] = 1.0
        return E

    Pq = np.zeros((nv, nv))
    q = np.zeros(nv)
    c0 = 0.0
    for i in range(N):
        Dx = sel(L.x(i)) - sel(L.xbar)
        Du = sel(L.u(i)) - sel(L.ubar)
        Pq += 2 * (Dx.T @ Q @ Dx + Du.T @ R @ Du)
    Dx = sel(L.x(N)) - sel(L.xbar)
    Pq += 2 * Dx.T @ Pm @ Dx
    Eb = sel(L.xbar)
    Pq += 2 * Eb.T @ T @ Eb
    q += -2 * Eb.T @ (T @ ref)
    c0 += float(ref @ T @ ref)

    Aeq, beq, G, h = [], [], [], []
    if received:
        HZ, hZ = np.asarray(p["HZW"], dtype=np.float64), np.asarray(p["hZW"], dtype=np.float64)
        G.append(-HZ @ sel(L.x(0)))
        h.append(hZ - HZ @ x_k)
    elif int(p["fixed_x0"]):
        Aeq.append(sel(L.x(0)))
        beq.append(x_k)
    else:
        HZ, hZ = np.asarray(p["HZ"], dtype=np.float64), np.asarray(p["hZ"], dtype=np.float64)
        G.append(-HZ @ sel(L.x(0)))
        h.append(hZ - HZ @ x_k)
    for i in range(N):
        Aeq.append(sel(L.x(i + 1)) - A @ sel(L.x(i)) - B @ sel(L.u(i)))
        beq.append(np.zeros(nx))
        G.append(Hx @ sel(L.x(i)))
        h.append(hx)
        G.append(Hu @ sel(L.u(i)))
        h.append(hu)
    Aeq.append((A - np.eye(nx)) @ sel(L.xbar) + B @ sel(L.ubar))
    beq.append(np.zeros(nx))
    if not received and int(p.get("terminal_equality", 0)):
        # TrackingMPC.py:105-107: the tracking MPC without a terminal set constrains x_N == x_bar
        Aeq.append(sel(L.x(N)) - sel(L.xbar))
        beq.append(np.zeros(nx))
    if projected:
        HTP, hTP = np.asarray(p["HTP"], dtype=np.float64), np.asarray(p["hTP"], dtype=np.float64)
        G.append(HTP[:, :nx] @ sel(L.xbar) + HTP[:, nx:] @ sel(L.ubar))
        h.append(hTP)
    elif received:
        G.append(HT[:, :nx] @ sel(L.xaux) + HT[:, nx:2 * nx] @ sel(L.xbar) + HT[:, 2 * nx:] @ sel(L.uaux))
        h.append(hT)
    else:
        G.append(HT[:, :nx] @ sel(L.x(N)) + HT[:, nx:2 * nx] @ sel(L.xbar) + HT[:, 2 * nx:] @ sel(L.ubar))
        h.append(hT)
    return dict(P=Pq, q=q, c0=c0, A=np.vstack(Aeq), b=np.concatenate(beq),
                G=np.vstack(G), h=np.concatenate(h), layout=L)


def unpack(qp: dict, v: np.ndarray):
    """-> x_nom (N+1,nx), u_nom (N,nu), x_ss (nx), u_ss (nu)."""
    L = qp["layout"]
    x = v[L.ox:L.ou].reshape(L.N + 1, L.nx)
    u = v[L.ou:L.oxb].reshape(L.N, L.nu)
    return x, u, v[L.xbar], v[L.ubar]


def pack(qp: dict, x_nom, u_nom, x_ss, u_ss):
    L = qp["layout"]
    v = np.zeros(L.nvar)
    v[L.ox:L.ou] = np.asarray(x_nom).reshape(-1)
    v[L.ou:L.oxb] = np.asarray(u_nom).reshape(-1)
    v[L.xbar] = x_ss
    v[L.ubar] = u_ss
    return v


def objective(qp: dict, v: np.ndarray) -> float:
    return float(0.5 * v @ qp["P"] @ v + qp["q"] @ v + qp["c0"])


def kkt_certificate(qp: dict, v: np.ndarray, act_tol: float = 1e-7):
    """Independent optimality certificate for a candidate primal point `v`.

    Identifies the active inequalities, recovers multipliers by a bounded
    least-squares fit of the stationarity equation, and reports

        r_stat  = || P v + q + A' y + G_act' lam ||_inf   (relative to max(1,|q|))
        r_eq    = || A v - b ||_inf
        r_ineq  = max(G v - h, 0)
        min_lam = smallest active multiplier (>= 0 required)

    Small values of the first three with min_lam >= 0 prove that `v` is the
    (unique, strictly convex case) minimiser, whichever solver produced it."""
    from scipy.optimize import lsq_linear
    P, q, A, b, G, h = (qp[k] for k in ("P", "q", "A", "b", "G", "h"))
    slack = h - G @ v
    scale = np.maximum(1.0, np.abs(h))
    act = np.flatnonzero(slack <= act_tol * scale)
    g = P @ v + q
    Mx = np.c_[A.T, G[act].T]
    lb = np.r_[np.full(A.shape[0], -np.inf), np.zeros(len(act))]
    colscale = np.maximum(np.linalg.norm(Mx, axis=0), 1e-300)
    res = lsq_linear(Mx / colscale, -g, bounds=(lb, np.full(Mx.shape[1], np.inf)),
                     method="bvls", tol=1e-15, max_iter=10 * Mx.shape[1] + 100)
    mult = res.x / colscale
    r = g + Mx @ mult
    lam = mult[A.shape[0]:]
    return dict(r_stat=float(np.max(np.abs(r)) / max(1.0, np.max(np.abs(q)))),
                r_eq=float(np.max(np.abs(A @ v - b))) if A.size else 0.0,
                r_ineq=float(max(0.0, np.max(-slack))),
                min_lam=float(lam.min()) if lam.size else 0.0,
                n_active=int(len(act)), active=act, lam=lam)


class SparseTemplate:
    """The un-condensed QP of one problem/variant with its parameter dependence factored out:
    P, A, G are constant; q = q0 + Qr ref, b = b0 + Bx x_k, h = h0 + Hx x_k (all affine in the
    two parameters of TubeTrackingMPC.py:121-122).  Built by 2 nx + 1 calls of build_sparse_qp,
    so it inherits that function's line-by-line correspondence with the reference."""

    def __init__(self, p: dict, variant: int = 0):
        nx = int(p["nx"])
        z = np.zeros(nx)
        base = build_sparse_qp(p, z, z, variant)
        self.P, self.A, self.G, self.layout = base["P"], base["A"], base["G"], base["layout"]
        self.q0, self.b0, self.h0 = base["q"], base["b"], base["h"]
        self.Qr = np.zeros((len(self.q0), nx))
        self.Bx = np.zeros((len(self.b0), nx))
        self.Hx = np.zeros((len(self.h0), nx))
        self.T = np.asarray(p["T"], dtype=np.float64)
        for j in range(nx):
            e = np.zeros(nx)
            e[j] = 1.0
            qx = build_sparse_qp(p, e, z, variant)
            self.Bx[:, j] = qx["b"] - self.b0
            self.Hx[:, j] = qx["h"] - self.h0
            self.Qr[:, j] = build_sparse_qp(p, z, e, variant)["q"] - self.q0

    def instance(self, x_k, ref) -> dict:
        x_k = np.asarray(x_k, dtype=np.float64).reshape(-1)
        ref = np.asarray(ref, dtype=np.float64).reshape(-1)
        return dict(P=self.P, q=self.q0 + self.Qr @ ref, c0=float(ref @ self.T @ ref), A=self.A, b=self.b0 + self.Bx @ x_k,
                    G=self.G, h=self.h0 + self.Hx @ x_k, layout=self.layout)


def kkt_certificate_fast(qp: dict, v: np.ndarray, act_tol: float = 1e-7):
    """kkt_certificate with the multipliers recovered by an unconstrained least-squares fit on the
    active rows first (degenerate vertices: minimum-norm solution); the bounded fit of
    kkt_certificate is the fall-back when that fit has a negative multiplier or a residual."""
    P, q, A, b, G, h = (qp[k] for k in ("P", "q", "A", "b", "G", "h"))
    slack = h - G @ v
    scale = np.maximum(1.0, np.abs(h))
    act = np.flatnonzero(slack <= act_tol * scale)
    g = P @ v + q
    Mx = np.c_[A.T, G[act].T]
    colscale = np.maximum(np.linalg.norm(Mx, axis=0), 1e-300)
    sol = np.linalg.lstsq(Mx / colscale, -g, rcond=1e-13)[0] / colscale
    lam = sol[A.shape[0]:]
    r = g + Mx @ sol
    qn = max(1.0, np.max(np.abs(q)))
    if (lam.size == 0 or lam.min() >= -1e-9 * max(1.0, np.abs(lam).max())) and np.max(np.abs(r)) <= 1e-7 * qn:
        return dict(r_stat=float(np.max(np.abs(r)) / qn),
                    r_eq=float(np.max(np.abs(A @ v - b))) if A.size else 0.0,
                    r_ineq=float(max(0.0, np.max(-slack))),
                    min_lam=float(max(lam.min(), 0.0)) if lam.size else 0.0,      # within -1e-9 |lam|_max of zero: counted as zero
                    n_active=int(len(act)), active=act, lam=lam)
    return kkt_certificate(qp, v, act_tol)


def _minimiser_on_set(qp: dict, v: np.ndarray, active=None, act_tol: float = 1e-7):
    """How far is the candidate `v` from THE minimiser, entry by entry?  Exact, not a residual norm.

    With the active set W identified at `v` (rows with slack <= act_tol; or handed in), the minimiser v_W of the
    equality-constrained QP  min 1/2 v'Pv + q'v  s.t.  A v = b, G_W v = h_W  is the solution of a LINEAR system;
    it is computed here by the null-space method, which tolerates the dependent rows of a degenerate vertex:

        C = [A; G_W],  c = [b; h_W],  v_p = v + C^+ (c - C v),  Z = null(C),
        (Z'PZ) w = -Z'(P v_p + q),    v_W = v_p + Z w.

    If v_W satisfies every other inequality and its multipliers on W (least-squares fit of C' mu = -(P v_W + q)) are
    non-negative, v_W is the minimiser of the inequality-constrained QP as well (strictly convex: unique), and
    `v_W - v` is the error of the candidate.  A residual-based certificate scales with |q| (1e6 for the cart-pole, whose
    terminal weight is 10 P, TubeTrackingMPC.py:27) and lets a 1e-6 shift of u_0 through; this does not.

    Returns dict(dv = v_W - v, du0 = max |dv| over the entries of u_0, certified = v_W passes the two checks and the
    face minimiser is resolved to 1e-11, r_ineq, min_mu, r_stat of v_W, resolution)."""
    P, q, A, b, G, h = (qp[k] for k in ("P", "q", "A", "b", "G", "h"))
    L = qp["layout"]
    if active is None:
        slack = h - G @ v
        active = np.flatnonzero(slack <= act_tol * np.maximum(1.0, np.abs(h)))
    C = np.vstack([A, G[active]]) if A.size else G[active]
    c = np.concatenate([b, h[active]]) if A.size else h[active]
    n = len(v)
    if C.shape[0]:
        rs = np.maximum(np.linalg.norm(C, axis=1), 1e-300)            # unit rows: the rank decision is about directions
        U, sv, Vt = np.linalg.svd(C / rs[:, None], full_matrices=True)
        rank = int((sv > 1e-10 * sv[0]).sum())
        Z = Vt[rank:].T
        resid = (c - C @ v) / rs
        v_p = v + Vt[:rank].T @ ((U[:, :rank].T @ resid) / sv[:rank])
    else:
        Z, v_p = np.eye(n), v.copy()
    last = 0.0
    v_w = v_p
    if Z.shape[1]:
        # Newton's step on the face is exact in exact arithmetic; |q| ~ 1e6 and cond(Z'PZ) ~ 1e5 leave 1e-9 .. 1e-8 of it
        # in floating point, so the step is repeated on what is left (iterative refinement); `last` = what the final
        # repetition still moved, the resolution of the distance returned
        Hz = Z.T @ P @ Z
        cho = np.linalg.cholesky(Hz)
        for _ in range(4):
            w = np.linalg.solve(cho.T, np.linalg.solve(cho, -Z.T @ (P @ v_w + q)))
            v_w = v_w + Z @ w
            last = float(np.max(np.abs(Z @ w)))
    g = P @ v_w + q
    if C.shape[0]:
        mu = (U[:, :rank] @ ((Vt[:rank] @ -g) / sv[:rank])) / rs      # minimum-norm multipliers, same rank decision as Z
        r_stat = float(np.max(np.abs(g + C.T @ mu)) / max(1.0, np.max(np.abs(q))))
        mu_in = mu[A.shape[0]:] if A.size else mu
        if len(mu_in) and mu_in.min() < 0.0:
            # degenerate vertex: the minimum-norm fit may put a negative weight on a dependent row although a
            # non-negative combination exists; ask for one
            from scipy.optimize import lsq_linear
            lb = np.r_[np.full(A.shape[0] if A.size else 0, -np.inf), np.zeros(len(active))]
            cs = np.maximum(np.linalg.norm(C, axis=1), 1e-300)
            res = lsq_linear(C.T / cs, -g, bounds=(lb, np.full(C.shape[0], np.inf)), method="bvls", tol=1e-15,
                             max_iter=10 * C.shape[0] + 100)
            mu2 = res.x / cs
            if np.max(np.abs(g + C.T @ mu2)) <= 1e-7 * max(1.0, np.max(np.abs(q))):
                mu = mu2
                r_stat = float(np.max(np.abs(g + C.T @ mu)) / max(1.0, np.max(np.abs(q))))
                mu_in = mu[A.shape[0]:] if A.size else mu
        mu_scale = max(1.0, float(np.abs(mu_in).max())) if len(mu_in) else 1.0
        min_mu = float(mu_in.min()) if len(mu_in) else 0.0
    else:
        r_stat, min_mu, mu_scale, mu_in = float(np.max(np.abs(g)) / max(1.0, np.max(np.abs(q)))), 0.0, 1.0, np.zeros(0)
    # violation of the other rows as a DISTANCE (rows of the un-condensed QP are not normalised: those of Z (-) W reach
    # norms of 1e3, and 1e-9 on such a row is 1e-12 in the variables)
    r_ineq = float(max(0.0, np.max((G @ v_w - h) / np.maximum(1.0, np.linalg.norm(G, axis=1))))) if G.size else 0.0
    dv = v_w - v
    return dict(dv=dv, du0=float(np.max(np.abs(dv[L.u(0)]))), r_ineq=r_ineq, min_mu=min_mu, r_stat=r_stat, mu_in=mu_in,
                resolution=last,
                certified=bool(r_ineq <= 1e-9 and min_mu >= -1e-9 * mu_scale and r_stat <= 1e-7 and last <= 1e-11),
                n_active=int(len(active)))


def minimiser_distance(qp: dict, v: np.ndarray, active=None, act_tol: float = 1e-7, rounds: int = 8):
    """`_minimiser_on_set` with the working set corrected when the one identified at `v` does not certify (a row within
    act_tol of its bound that is not active at the minimiser, or the reverse): the row with the most negative multiplier
    leaves, violated rows enter -- a textbook primal-dual active-set correction, a few rounds at most because `v` is
    already close.  The returned distance refers to the certified minimiser; `certified` False means no working set
    certified within `rounds`, and the distance says nothing."""
    G, h, A = qp["G"], qp["h"], qp["A"]
    if active is None:
        active = np.flatnonzero(h - G @ v <= act_tol * np.maximum(1.0, np.abs(h)))
    active = np.asarray(active, dtype=int)
    rn = np.maximum(1.0, np.linalg.norm(G, axis=1))
    slack0 = (h - G @ v) / rn
    d = None
    for _ in range(rounds):
        d = _minimiser_on_set(qp, v, active)
        if d["certified"]:
            break
        v_w = v + d["dv"]
        # a working set whose rows cannot all sit on their bounds (nearly parallel facets of the 854-row initial-state set
        # with different offsets: the dependent system is met in the least-squares sense only): the row that was furthest
        # from its bound at `v` leaves
        loose = np.abs(G[active] @ v_w - h[active]) / rn[active] > 1e-9
        if loose.any():
            cand = active[loose]
            active = np.setdiff1d(active, [cand[int(np.argmax(slack0[cand]))]])
            continue
        viol = np.flatnonzero((G @ v_w - h) / np.maximum(1.0, np.linalg.norm(G, axis=1)) > 1e-9)
        viol = np.setdiff1d(viol, active)
        if len(viol):
            active = np.union1d(active, viol)
            continue
        mu_in = d["mu_in"]
        if len(mu_in) and mu_in.min() < 0.0:
            active = np.delete(active, int(np.argmin(mu_in)))
            continue
        break
    if d is not None and d["certified"] and len(d["mu_in"]) and float(np.max(np.abs(d["dv"]))) > 1e-9:
        # A row within act_tol of its bound that is NOT active at the minimiser, kept in the set with a zero multiplier (the
        # bounded least-squares fit above puts it there): stationarity holds only to the certificate's 1e-7 |q| -- 0.1 in
        # absolute terms for the cart-pole -- and the point found is the minimiser on too small a face (seen: u_2 off by
        # 0.1 with a HIGHER objective than the candidate).  Rows that carry no multiplier are not needed to hold the
        # minimiser: without them the face is larger, and if its minimiser certifies with an objective no higher, it is
        # the better answer.
        mu_in = d["mu_in"]
        keep = mu_in > 1e-9 * max(1.0, float(np.abs(mu_in).max()))
        if not keep.all():
            d2 = _minimiser_on_set(qp, v, active[keep])
            if d2["certified"] and objective(qp, v + d2["dv"]) <= objective(qp, v + d["dv"]):
                d, active = d2, active[keep]
    d["active"] = active
    return d


def lp_infeasible(qp: dict) -> bool:
    """True iff {A v = b, G v <= h} is empty according to HiGHS (scipy.optimize.linprog, the
    LP solver the reference itself uses, utils_polytope.py:19) -- the solver-independent
    counterpart of a TMPC_STATUS_INFEASIBLE answer."""
    from scipy.optimize import linprog
    n = qp["G"].shape[1]
    res = linprog(np.zeros(n), A_ub=qp["G"], b_ub=qp["h"], A_eq=qp["A"] if qp["A"].size else None,
                  b_eq=qp["b"] if qp["A"].size else None, bounds=[(None, None)] * n, method="highs")
    return res.status == 2
