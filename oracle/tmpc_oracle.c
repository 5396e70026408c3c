/*
 * tmpc_oracle.c -- ORACLE.  Test infrastructure, NOT product code.
 *
 * CPU restatement (plain C, float64) of the per-timestep tube-tracking QP of
 * EricssonResearch/Robust-Tracking-MPC-over-Lossy-Networks.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library
 * built from this file; the product path (libtmpc_hip.so) never does.
 *
 * PARITY UNPINNED: the reference holds no golden vectors / known-answer tests for
 * the QP solution, and its solver stack (cvxpy -> Clarabel, un-pinned,
 * setup.cfg:13-20) cannot be installed here.  What pins this file instead is
 * tests/test_oracle.py: an independent KKT certificate (numpy) of its outputs, the
 * numpy restatement oracle/qp_sparse.py + oracle/ipm_numpy.py, and a scipy
 * cross-check at small sizes.
 *
 * What is restated, and from where (reference src/LinearMPCOverNetworks/):
 *
 *   build_sparse()   the QP exactly as TubeTrackingMPC.generate_optimization_problem
 *                    writes it for cvxpy (TubeTrackingMPC.py:104-156): variables
 *                    x_mpc, u_mpc, x_bar, u_bar (:117-120), initial-state equality
 *                    (:127) or tube inequality (:132), dynamics (:138), stage
 *                    constraints (:139-140), steady-state equality (:147), terminal
 *                    inequality (:149), cost (:136,:143,:144); and the packet-received
 *                    problem of ExtendedTubeTrackingMPC (:253-299) with its terminal
 *                    row on the base problem's x_mpc[:,N] / u_bar (:293), which are
 *                    free auxiliary variables there.
 *   solve_dense()    the solve the reference delegates to Clarabel (:183, :320, :337).
 *                    Clarabel is an interior-point method; its source is not in the
 *                    reference tree.  Here: the equalities are removed with an
 *                    orthonormal null-space basis (Householder QR -- purely numerical,
 *                    deliberately NOT the prediction-matrix condensing the HIP library
 *                    uses, so that one checks the other), then a Mehrotra
 *                    predictor-corrector interior-point iteration, then an active-set
 *                    refinement that takes the iterate to the exact minimiser.  The
 *                    refinement matters: the cost mixes weights of 1e5..1e6 (P, T) with
 *                    R = 0.1, so stopping at the reference's own gap tolerance of 1e-7
 *                    leaves u_0 uncertain by 1e-3 or more.
 *
 * The minimiser is unique (strictly convex cost in the free directions), so an exact
 * solve of the same QP is what any correct solver converges to.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/tmpc.h"

/* ------------------------------------------------------------------ helpers */
static double *dalloc(size_t n) { return (double *)calloc(n ? n : 1, sizeof(double)); }

/* C(m x n) = A(m x k) * B(k x n) */
static void matmul(const double *A, const double *B, double *C, int m, int k, int n) {
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0;
            for (int l = 0; l < k; ++l) s += A[i * k + l] * B[l * n + j];
            C[i * n + j] = s;
        }
}
/* C(m x n) = A^T * B, A is k x m, B is k x n */
static void matmul_tn(const double *A, const double *B, double *C, int k, int m, int n) {
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0;
            for (int l = 0; l < k; ++l) s += A[l * m + i] * B[l * n + j];
            C[i * n + j] = s;
        }
}
/* in-place lower Cholesky of the n x n matrix M (row-major, lower part used).
 * returns 0 on success, j+1 if pivot j is not positive */
static int chol(double *M, int n) {
    for (int j = 0; j < n; ++j) {
        double v = M[j * n + j];
        for (int k = 0; k < j; ++k) v -= M[j * n + k] * M[j * n + k];
        if (!(v > 0)) return j + 1;
        v = sqrt(v);
        M[j * n + j] = v;
        for (int i = j + 1; i < n; ++i) {
            double t = M[i * n + j];
            for (int k = 0; k < j; ++k) t -= M[i * n + k] * M[j * n + k];
            M[i * n + j] = t / v;
        }
    }
    return 0;
}
static void chol_solve(const double *L, int n, double *b) {
    for (int i = 0; i < n; ++i) {
        double t = b[i];
        for (int k = 0; k < i; ++k) t -= L[i * n + k] * b[k];
        b[i] = t / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double t = b[i];
        for (int k = i + 1; k < n; ++k) t -= L[k * n + i] * b[k];
        b[i] = t / L[i * n + i];
    }
}

/* ------------------------------------------------------------------ data */
typedef struct {
    int nvar, nw, nc, npar, nx;
    /* v = V0 x_k + Zb w */
    double *V0, *Zb;
    /* reduced QP: min 1/2 w'Hw + (F1 x + F2 r)'w  s.t. G w <= g0 + E x  (scaled copies below) */
    double *H, *F1, *F2, *G, *g0, *E;
    /* rows that depend on x_k only:  0 <= gp0 + Ep x */
    double *gp0, *Ep;
    /* scaling: w = Dv ws ; rows divided by rn */
    double *Dv, *Hs, *Hinv, *Gs, *g0s, *Es, *F1s, *F2s;
    int always_infeasible;
} form_t;

struct oracle_handle {
    int nx, nu, N, nvariants, max_iter;
    double tol;
    form_t f[2];
    char err[256];
};
typedef struct oracle_handle oracle_handle;

static char g_err[256];

/* ------------------------------------------------------------------ sparse form */
typedef struct {
    int nvar, me, mi;
    double *P, *Qr;        /* cost: 1/2 v'Pv + (Qr ref)'v            Qr: nvar x nx */
    double *Aeq, *Beq;     /* Aeq v = Beq x_k                         Beq: me x nx  */
    double *G, *h0, *Eh;   /* G v <= h0 + Eh x_k                      Eh: mi x nx   */
} sparse_t;

/* adds w * D' W D to P where D = sel(a) - sel(b): blocks of size n, a/b = start columns */
static void add_diff_quad(double *P, int nvar, int a, int b, const double *W, int n, double w) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double v = w * W[i * n + j];
            P[(a + i) * nvar + a + j] += v;
            P[(b + i) * nvar + b + j] += v;
            P[(a + i) * nvar + b + j] -= v;
            P[(b + i) * nvar + a + j] -= v;
        }
}

static int build_sparse(const tmpc_problem *p, int variant, sparse_t *s) {
    const int nx = p->nx, nu = p->nu, N = p->N;
    const int received = (variant == 1);
    /* :293 literally = free auxiliaries; eliminated when the projection (HTP, hTP) is supplied (include/tmpc.h) */
    const int projected = received && p->literal_terminal_row && p->rTP > 0 && p->HTP && p->hTP;
    const int aux = received && p->literal_terminal_row && !projected;
    const int ox = 0, ou = nx * (N + 1), oxb = ou + nu * N, oub = oxb + nx, oxa = oub + nu, oua = oxa + nx;
    const int nvar = oub + nu + (aux ? nx + nu : 0);
    const int fixed = (!received) && p->fixed_x0;
    const int rz = received ? p->rZW : (fixed ? 0 : p->rZ);
    const double *HZ = received ? p->HZW : p->HZ, *hZ = received ? p->hZW : p->hZ;
    if (rz > 0 && (!HZ || !hZ)) return -1;
    /* TrackingMPC.py:105-107: without a terminal set the tracking MPC constrains x_N == x_bar */
    const int term_eq = (!received) && p->terminal_equality;
    if (term_eq && p->rT > 0) return -1;
    const int me = (fixed ? nx : 0) + nx * N + nx + (term_eq ? nx : 0);
    const int mi = rz + N * (p->rx + p->ru) + (projected ? p->rTP : p->rT);
    s->nvar = nvar; s->me = me; s->mi = mi;
    s->P = dalloc((size_t)nvar * nvar); s->Qr = dalloc((size_t)nvar * nx);
    s->Aeq = dalloc((size_t)me * nvar); s->Beq = dalloc((size_t)me * nx);
    s->G = dalloc((size_t)mi * nvar); s->h0 = dalloc(mi); s->Eh = dalloc((size_t)mi * nx);
    /* cost (TubeTrackingMPC.py:136,143,144); factor 2 because the form is 1/2 v'Pv */
    for (int i = 0; i < N; ++i) {
        add_diff_quad(s->P, nvar, ox + i * nx, oxb, p->Q, nx, 2.0);
        add_diff_quad(s->P, nvar, ou + i * nu, oub, p->R, nu, 2.0);
    }
    add_diff_quad(s->P, nvar, ox + N * nx, oxb, p->P, nx, 2.0);
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < nx; ++j) {
            s->P[(oxb + i) * nvar + oxb + j] += 2.0 * p->T[i * nx + j];
            s->Qr[(oxb + i) * nx + j] = -2.0 * p->T[i * nx + j];   /* -2 T ref on x_bar */
        }
    if (aux) {
        /* cost-free auxiliaries (:293): vanishing weight eps = 2e-6 min(diag R) so that the
         * problem is strictly convex; same rule as the HIP library */
        double rmin = p->R[0];
        for (int i = 1; i < nu; ++i) if (p->R[i * nu + i] < rmin) rmin = p->R[i * nu + i];
        for (int i = oxa; i < nvar; ++i) s->P[i * nvar + i] += 2e-6 * rmin;
    }
    int er = 0, ir = 0;
    /* initial state */
    if (fixed) {
        for (int i = 0; i < nx; ++i) { s->Aeq[(er + i) * nvar + ox + i] = 1.0; s->Beq[(er + i) * nx + i] = 1.0; }
        er += nx;
    } else {
        /* Hz (x_k - x_0) <= hz   ->   -Hz x_0 <= hz - Hz x_k */
        for (int r = 0; r < rz; ++r) {
            for (int j = 0; j < nx; ++j) {
                s->G[(ir + r) * nvar + ox + j] = -HZ[r * nx + j];
                s->Eh[(ir + r) * nx + j] = -HZ[r * nx + j];
            }
            s->h0[ir + r] = hZ[r];
        }
        ir += rz;
    }
    for (int i = 0; i < N; ++i) {
        /* x_{i+1} - A x_i - B u_i = 0 */
        for (int r = 0; r < nx; ++r) {
            s->Aeq[(er + r) * nvar + ox + (i + 1) * nx + r] = 1.0;
            for (int j = 0; j < nx; ++j) s->Aeq[(er + r) * nvar + ox + i * nx + j] -= p->A[r * nx + j];
            for (int j = 0; j < nu; ++j) s->Aeq[(er + r) * nvar + ou + i * nu + j] -= p->B[r * nu + j];
        }
        er += nx;
        for (int r = 0; r < p->rx; ++r) {
            for (int j = 0; j < nx; ++j) s->G[(ir + r) * nvar + ox + i * nx + j] = p->Hx[r * nx + j];
            s->h0[ir + r] = p->hx[r];
        }
        ir += p->rx;
        for (int r = 0; r < p->ru; ++r) {
            for (int j = 0; j < nu; ++j) s->G[(ir + r) * nvar + ou + i * nu + j] = p->Hu[r * nu + j];
            s->h0[ir + r] = p->hu[r];
        }
        ir += p->ru;
    }
    /* (A - I) x_bar + B u_bar = 0 */
    for (int r = 0; r < nx; ++r) {
        for (int j = 0; j < nx; ++j) s->Aeq[(er + r) * nvar + oxb + j] = p->A[r * nx + j] - (r == j ? 1.0 : 0.0);
        for (int j = 0; j < nu; ++j) s->Aeq[(er + r) * nvar + oub + j] = p->B[r * nu + j];
    }
    er += nx;
    if (term_eq) {
        /* x_N - x_bar = 0 */
        for (int r = 0; r < nx; ++r) {
            s->Aeq[(er + r) * nvar + ox + N * nx + r] = 1.0;
            s->Aeq[(er + r) * nvar + oxb + r] = -1.0;
        }
        er += nx;
    }
    /* terminal */
    if (projected) {
        for (int r = 0; r < p->rTP; ++r) {
            for (int j = 0; j < nx; ++j) s->G[(ir + r) * nvar + oxb + j] += p->HTP[r * (nx + nu) + j];
            for (int j = 0; j < nu; ++j) s->G[(ir + r) * nvar + oub + j] += p->HTP[r * (nx + nu) + nx + j];
            s->h0[ir + r] = p->hTP[r];
        }
        ir += p->rTP;
    } else {
        const int cx = aux ? oxa : ox + N * nx, cu = aux ? oua : oub, w = 2 * nx + nu;
        for (int r = 0; r < p->rT; ++r) {
            for (int j = 0; j < nx; ++j) s->G[(ir + r) * nvar + cx + j] += p->HT[r * w + j];
            for (int j = 0; j < nx; ++j) s->G[(ir + r) * nvar + oxb + j] += p->HT[r * w + nx + j];
            for (int j = 0; j < nu; ++j) s->G[(ir + r) * nvar + cu + j] += p->HT[r * w + 2 * nx + j];
            s->h0[ir + r] = p->hT[r];
        }
        ir += p->rT;
    }
    return (er == me && ir == mi) ? 0 : -2;
}

static void free_sparse(sparse_t *s) {
    free(s->P); free(s->Qr); free(s->Aeq); free(s->Beq); free(s->G); free(s->h0); free(s->Eh);
}

/* Householder QR of M (m x n, m >= n, row-major, overwritten by R in its upper
 * triangle); Q (m x m) is formed explicitly. Returns 0, or -1 if rank deficient. */
static int householder_qr(double *M, int m, int n, double *Q) {
    memset(Q, 0, sizeof(double) * m * m);
    for (int i = 0; i < m; ++i) Q[i * m + i] = 1.0;
    double *v = dalloc(m);
    int rc = 0;
    for (int k = 0; k < n; ++k) {
        double nrm = 0;
        for (int i = k; i < m; ++i) nrm += M[i * n + k] * M[i * n + k];
        nrm = sqrt(nrm);
        if (nrm < 1e-13) { rc = -1; continue; }
        double alpha = M[k * n + k] > 0 ? -nrm : nrm;
        for (int i = 0; i < m; ++i) v[i] = 0;
        for (int i = k; i < m; ++i) v[i] = M[i * n + k];
        v[k] -= alpha;
        double vn = 0;
        for (int i = k; i < m; ++i) vn += v[i] * v[i];
        if (vn < 1e-300) continue;
        for (int j = k; j < n; ++j) {        /* M <- (I - 2 v v'/v'v) M */
            double d = 0;
            for (int i = k; i < m; ++i) d += v[i] * M[i * n + j];
            d = 2 * d / vn;
            for (int i = k; i < m; ++i) M[i * n + j] -= d * v[i];
        }
        for (int i = 0; i < m; ++i) {        /* Q <- Q (I - 2 v v'/v'v) */
            double d = 0;
            for (int j = k; j < m; ++j) d += Q[i * m + j] * v[j];
            d = 2 * d / vn;
            for (int j = k; j < m; ++j) Q[i * m + j] -= d * v[j];
        }
    }
    free(v);
    return rc;
}

/* ------------------------------------------------------------------ reduction + scaling */
static int build_form(const tmpc_problem *p, int variant, form_t *f) {
    sparse_t s;
    memset(&s, 0, sizeof(s));
    int rc = build_sparse(p, variant, &s);
    if (rc) { free_sparse(&s); return rc; }
    const int nvar = s.nvar, me = s.me, mi = s.mi, nx = p->nx, nw = nvar - me;
    /* QR of Aeq^T */
    double *At = dalloc((size_t)nvar * me), *Q = dalloc((size_t)nvar * nvar);
    for (int i = 0; i < me; ++i) for (int j = 0; j < nvar; ++j) At[j * me + i] = s.Aeq[i * nvar + j];
    if (householder_qr(At, nvar, me, Q)) { free(At); free(Q); free_sparse(&s); return -3; }
    /* V0 = Y R^-T Beq : solve R^T t = Beq (R upper me x me in At), V0 = Y t */
    double *t = dalloc((size_t)me * nx);
    for (int c = 0; c < nx; ++c)
        for (int i = 0; i < me; ++i) {
            double v = s.Beq[i * nx + c];
            for (int k = 0; k < i; ++k) v -= At[k * me + i] * t[k * nx + c];
            t[i * nx + c] = v / At[i * me + i];
        }
    f->nvar = nvar; f->nw = nw; f->nx = nx;
    f->V0 = dalloc((size_t)nvar * nx); f->Zb = dalloc((size_t)nvar * nw);
    for (int i = 0; i < nvar; ++i) {
        for (int c = 0; c < nx; ++c) {
            double v = 0;
            for (int k = 0; k < me; ++k) v += Q[i * nvar + k] * t[k * nx + c];
            f->V0[i * nx + c] = v;
        }
        for (int j = 0; j < nw; ++j) f->Zb[i * nw + j] = Q[i * nvar + me + j];
    }
    /* reduced blocks */
    double *PZ = dalloc((size_t)nvar * nw), *PV = dalloc((size_t)nvar * nx);
    double *H = dalloc((size_t)nw * nw), *F1 = dalloc((size_t)nw * nx), *F2 = dalloc((size_t)nw * nx);
    double *Gr = dalloc((size_t)mi * nw), *GV = dalloc((size_t)mi * nx), *E = dalloc((size_t)mi * nx);
    matmul(s.P, f->Zb, PZ, nvar, nvar, nw);
    matmul_tn(f->Zb, PZ, H, nvar, nw, nw);
    matmul(s.P, f->V0, PV, nvar, nvar, nx);
    matmul_tn(f->Zb, PV, F1, nvar, nw, nx);
    matmul_tn(f->Zb, s.Qr, F2, nvar, nw, nx);
    matmul(s.G, f->Zb, Gr, mi, nvar, nw);
    matmul(s.G, f->V0, GV, mi, nvar, nx);
    for (int i = 0; i < mi * nx; ++i) E[i] = s.Eh[i] - GV[i];
    for (int i = 0; i < nw; ++i) for (int j = 0; j < i; ++j) { double a = 0.5 * (H[i * nw + j] + H[j * nw + i]); H[i * nw + j] = H[j * nw + i] = a; }
    /* row classification */
    int nc = 0, npar = 0;
    int *kind = (int *)calloc(mi ? mi : 1, sizeof(int));   /* 0 keep, 1 x-only, 2 drop */
    f->always_infeasible = 0;
    for (int r = 0; r < mi; ++r) {
        double gn = 0, gn0 = 0, en = 0;
        for (int j = 0; j < nw; ++j) gn += Gr[r * nw + j] * Gr[r * nw + j];
        for (int j = 0; j < nvar; ++j) gn0 += s.G[r * nvar + j] * s.G[r * nvar + j];
        for (int j = 0; j < nx; ++j) en += E[r * nx + j] * E[r * nx + j];
        if (sqrt(gn) > 1e-11 * (1.0 + sqrt(gn0))) { kind[r] = 0; ++nc; }
        else if (sqrt(en) > 1e-11 * (1.0 + sqrt(gn0))) { kind[r] = 1; ++npar; }
        else { kind[r] = 2; if (s.h0[r] < -1e-9 * (1.0 + fabs(s.h0[r]))) f->always_infeasible = 1; }
    }
    f->nc = nc; f->npar = npar;
    f->H = H; f->F1 = F1; f->F2 = F2;
    f->G = dalloc((size_t)nc * nw); f->g0 = dalloc(nc); f->E = dalloc((size_t)nc * nx);
    f->gp0 = dalloc(npar); f->Ep = dalloc((size_t)npar * nx);
    for (int r = 0, a = 0, b = 0; r < mi; ++r) {
        if (kind[r] == 0) {
            memcpy(f->G + (size_t)a * nw, Gr + (size_t)r * nw, sizeof(double) * nw);
            memcpy(f->E + (size_t)a * nx, E + (size_t)r * nx, sizeof(double) * nx);
            f->g0[a++] = s.h0[r];
        } else if (kind[r] == 1) {
            memcpy(f->Ep + (size_t)b * nx, E + (size_t)r * nx, sizeof(double) * nx);
            f->gp0[b++] = s.h0[r];
        }
    }
    /* scaling: Jacobi on the variables, unit rows */
    f->Dv = dalloc(nw); f->Hs = dalloc((size_t)nw * nw); f->Hinv = dalloc((size_t)nw * nw);
    f->Gs = dalloc((size_t)nc * nw); f->g0s = dalloc(nc); f->Es = dalloc((size_t)nc * nx);
    f->F1s = dalloc((size_t)nw * nx); f->F2s = dalloc((size_t)nw * nx);
    for (int i = 0; i < nw; ++i) {
        if (!(H[i * nw + i] > 0)) { rc = -4; }
        f->Dv[i] = 1.0 / sqrt(fabs(H[i * nw + i]) + 1e-300);
    }
    for (int i = 0; i < nw; ++i) {
        for (int j = 0; j < nw; ++j) f->Hs[i * nw + j] = H[i * nw + j] * f->Dv[i] * f->Dv[j];
        for (int c = 0; c < nx; ++c) { f->F1s[i * nx + c] = F1[i * nx + c] * f->Dv[i]; f->F2s[i * nx + c] = F2[i * nx + c] * f->Dv[i]; }
    }
    for (int r = 0; r < nc; ++r) {
        double n2 = 0;
        for (int j = 0; j < nw; ++j) { double v = f->G[r * nw + j] * f->Dv[j]; f->Gs[r * nw + j] = v; n2 += v * v; }
        double rn = sqrt(n2);
        for (int j = 0; j < nw; ++j) f->Gs[r * nw + j] /= rn;
        f->g0s[r] = f->g0[r] / rn;
        for (int c = 0; c < nx; ++c) f->Es[r * nx + c] = f->E[r * nx + c] / rn;
    }
    /* Hinv = Hs^-1 through Cholesky */
    {
        double *L = dalloc((size_t)nw * nw), *col = dalloc(nw);
        memcpy(L, f->Hs, sizeof(double) * nw * nw);
        if (chol(L, nw)) rc = -4;
        else
            for (int c = 0; c < nw; ++c) {
                memset(col, 0, sizeof(double) * nw); col[c] = 1.0;
                chol_solve(L, nw, col);
                for (int i = 0; i < nw; ++i) f->Hinv[i * nw + c] = col[i];
            }
        free(L); free(col);
    }
    free(kind); free(At); free(Q); free(t); free(PZ); free(PV); free(Gr); free(GV); free(E);
    free_sparse(&s);
    return rc;
}

static void free_form(form_t *f) {
    free(f->V0); free(f->Zb); free(f->H); free(f->F1); free(f->F2); free(f->G); free(f->g0); free(f->E);
    free(f->gp0); free(f->Ep); free(f->Dv); free(f->Hs); free(f->Hinv); free(f->Gs); free(f->g0s); free(f->Es);
    free(f->F1s); free(f->F2s);
    memset(f, 0, sizeof(*f));
}

/* ------------------------------------------------------------------ dense solver */
typedef struct {
    double *q, *h, *z, *s, *lam, *rp, *d, *M, *L, *rhs, *dz, *ds, *dl, *dsa, *dla, *gz, *tmp;
    /* polish */
    int *W; unsigned char *inW;
    double *yall, *T, *S, *y, *zp, *r1, *r2, *t1, *dy, *r;
    int wcap;
} work_t;

static work_t *work_alloc(int nv, int nc) {
    work_t *w = (work_t *)calloc(1, sizeof(work_t));
    w->q = dalloc(nv); w->h = dalloc(nc); w->z = dalloc(nv); w->s = dalloc(nc); w->lam = dalloc(nc);
    w->rp = dalloc(nc); w->d = dalloc(nc); w->M = dalloc((size_t)nv * nv); w->L = dalloc((size_t)nv * nv);
    w->rhs = dalloc(nv); w->dz = dalloc(nv); w->ds = dalloc(nc); w->dl = dalloc(nc); w->dsa = dalloc(nc);
    w->dla = dalloc(nc); w->gz = dalloc(nc); w->tmp = dalloc(nc > nv ? nc : nv);
    w->wcap = nc < 4 * nv + 32 ? nc : 4 * nv + 32;
    w->W = (int *)calloc(w->wcap + 1, sizeof(int)); w->inW = (unsigned char *)calloc(nc + 1, 1);
    w->yall = dalloc(nc); w->T = dalloc((size_t)nv * w->wcap); w->S = dalloc((size_t)w->wcap * w->wcap);
    w->y = dalloc(w->wcap); w->zp = dalloc(nv); w->r1 = dalloc(nv); w->r2 = dalloc(w->wcap); w->t1 = dalloc(nv);
    w->dy = dalloc(w->wcap); w->r = dalloc(nc);
    return w;
}
static void work_free(work_t *w) {
    free(w->q); free(w->h); free(w->z); free(w->s); free(w->lam); free(w->rp); free(w->d); free(w->M); free(w->L);
    free(w->rhs); free(w->dz); free(w->ds); free(w->dl); free(w->dsa); free(w->dla); free(w->gz); free(w->tmp);
    free(w->W); free(w->inW); free(w->yall); free(w->T); free(w->S); free(w->y); free(w->zp); free(w->r1);
    free(w->r2); free(w->t1); free(w->dy); free(w->r); free(w);
}

static double max_step(const double *s, const double *ds, const double *l, const double *dl, int n, double tau) {
    double a = 1.0;
    for (int i = 0; i < n; ++i) {
        if (ds[i] < 0) { double t = -tau * s[i] / ds[i]; if (t < a) a = t; }
        if (dl[i] < 0) { double t = -tau * l[i] / dl[i]; if (t < a) a = t; }
    }
    return a;
}

/* Active-set refinement ("polish") of an interior-point iterate (z, lam, s) of the
 * scaled QP.  W = {lam_i > s_i}.  On W the KKT system
 *       Hs z + q + G_W' y = 0,   G_W z = h_W
 * is solved by proximal Newton steps on its range-space form (S + delta I) dy = ...,
 * S = G_W Hs^-1 G_W', which tolerates linearly dependent rows of G_W (degenerate
 * vertices are the rule for this problem class) and keeps y next to the
 * interior-point multipliers.  The result is accepted only if it is primal
 * feasible on all rows and y >= 0; otherwise W is corrected and the step repeated.
 * Returns 1 on success (z, lam overwritten by the exact KKT point). */
/* developer statistics (single-threaded runs only): refinement calls / rounds / successes */
static long g_polish_calls = 0, g_polish_rounds = 0, g_polish_ok = 0;
void oracle_debug_counters(long *out3, int reset) {
    if (out3) { out3[0] = g_polish_calls; out3[1] = g_polish_rounds; out3[2] = g_polish_ok; }
    if (reset) g_polish_calls = g_polish_rounds = g_polish_ok = 0;
}
static int polish(const form_t *f, work_t *w) {
    ++g_polish_calls;
    const int nv = f->nw, nc = f->nc;
    const double *Gs = f->Gs, *Hs = f->Hs, *Hinv = f->Hinv;
    for (int i = 0; i < nc; ++i) { w->inW[i] = w->lam[i] > w->s[i]; w->yall[i] = w->lam[i]; }
    memcpy(w->zp, w->z, sizeof(double) * nv);
    int loose_retries = 0;      /* rounds that only repeat the Newton steps on an unchanged working set (see below) */
    for (int it = 0; it < 6 + loose_retries; ++it) {
        ++g_polish_rounds;
        int m = 0;
        for (int i = 0; i < nc; ++i) if (w->inW[i]) { if (m >= w->wcap) return 0; w->W[m++] = i; }
        if (m == 0) {
            for (int i = 0; i < nv; ++i) { double v = 0; for (int j = 0; j < nv; ++j) v -= Hinv[i * nv + j] * w->q[j]; w->zp[i] = v; }
        } else {
            /* T = Hinv G_W' (nv x m), S = G_W T */
            for (int i = 0; i < nv; ++i)
                for (int k = 0; k < m; ++k) {
                    const double *g = Gs + (size_t)w->W[k] * nv; double v = 0;
                    for (int j = 0; j < nv; ++j) v += Hinv[i * nv + j] * g[j];
                    w->T[i * m + k] = v;
                }
            double dmax = 0;
            for (int a = 0; a < m; ++a)
                for (int b = 0; b <= a; ++b) {
                    const double *g = Gs + (size_t)w->W[a] * nv; double v = 0;
                    for (int j = 0; j < nv; ++j) v += g[j] * w->T[j * m + b];
                    w->S[a * m + b] = v; w->S[b * m + a] = v;
                    if (a == b && v > dmax) dmax = v;
                }
            for (int a = 0; a < m; ++a) w->S[a * m + a] += 1e-11 * dmax;
            if (chol(w->S, m)) return 0;
            for (int k = 0; k < m; ++k) w->y[k] = w->yall[w->W[k]];
            for (int step = 0; step < (loose_retries ? 12 : 4); ++step) {
                for (int i = 0; i < nv; ++i) {           /* r1 = Hs z + q + G_W' y */
                    double v = w->q[i];
                    for (int j = 0; j < nv; ++j) v += Hs[i * nv + j] * w->zp[j];
                    for (int k = 0; k < m; ++k) v += Gs[(size_t)w->W[k] * nv + i] * w->y[k];
                    w->r1[i] = v;
                }
                if (getenv("ORACLE_DEBUG")) { double rn = 0; for (int i = 0; i < nv; ++i) if (fabs(w->r1[i]) > rn) rn = fabs(w->r1[i]); fprintf(stderr, "   polish it %d m %d step %d |r1| %.3e dmax %.3e\n", it, m, step, rn, dmax); }
                for (int i = 0; i < nv; ++i) { double v = 0; for (int j = 0; j < nv; ++j) v += Hinv[i * nv + j] * w->r1[j]; w->t1[i] = v; }
                for (int k = 0; k < m; ++k) {            /* dy rhs = r2 - G_W t1 */
                    const double *g = Gs + (size_t)w->W[k] * nv; double gz = 0, gt = 0;
                    for (int j = 0; j < nv; ++j) { gz += g[j] * w->zp[j]; gt += g[j] * w->t1[j]; }
                    w->dy[k] = (gz - w->h[w->W[k]]) - gt;
                }
                chol_solve(w->S, m, w->dy);
                double dzn = 0, zn = 1.0;
                for (int i = 0; i < nv; ++i) {
                    double v = w->t1[i];
                    for (int k = 0; k < m; ++k) v += w->T[i * m + k] * w->dy[k];
                    w->zp[i] -= v;
                    if (fabs(v) > dzn) dzn = fabs(v);
                    if (fabs(w->zp[i]) > zn) zn = fabs(w->zp[i]);
                }
                for (int k = 0; k < m; ++k) w->y[k] += w->dy[k];
                if (step >= 1 && dzn <= 1e-14 * zn) break;       /* the step no longer moves the iterate */
            }
        }
        /* verify */
        int nviol = 0, nneg = 0, nloose = 0;
        double ymax = 1.0;
        for (int k = 0; k < m; ++k) if (fabs(w->y[k]) > ymax) ymax = fabs(w->y[k]);
        for (int i = 0; i < nc; ++i) {
            const double *g = Gs + (size_t)i * nv; double v = -w->h[i];
            for (int j = 0; j < nv; ++j) v += g[j] * w->zp[j];
            w->r[i] = v;
            const double hi = fabs(w->h[i]) > 1.0 ? fabs(w->h[i]) : 1.0;
            if (!w->inW[i] && v > 1e-12 * hi) ++nviol;
            if (w->inW[i] && fabs(v) > 1e-11 * hi) ++nloose;    /* working-set row not on its bound: not converged */
        }
        for (int k = 0; k < m; ++k) if (w->y[k] < -1e-10 * ymax) ++nneg;
        if (getenv("ORACLE_DEBUG")) fprintf(stderr, "   polish it %d m %d nviol %d nneg %d nloose %d\n", it, m, nviol, nneg, nloose);
        /* rows of W off their bound with nothing left to correct: the steps have not converged, give up.
         * (With wrong rows still in W the system is inconsistent and looseness is expected: correct W first.) */
        if (nloose && nviol == 0 && nneg == 0) {
            /* Nearly parallel working rows (neighbouring facets of the 854-row initial-state set, both active): S is nearly
             * singular along their difference and the proximal steps contract that component slowly, although z no longer
             * moves.  The working set is right: the same set gets up to two more rounds of twelve steps before giving up. */
            if (loose_retries >= 2) return 0;
            ++loose_retries;
            for (int k = 0; k < m; ++k) w->yall[w->W[k]] = w->y[k];
            continue;
        }
        if (nviol == 0 && nneg == 0) {
            memcpy(w->z, w->zp, sizeof(double) * nv);
            for (int i = 0; i < nc; ++i) { w->lam[i] = 0; w->s[i] = w->r[i] < 0 ? -w->r[i] : 0; }
            for (int k = 0; k < m; ++k) w->lam[w->W[k]] = w->y[k] > 0 ? w->y[k] : 0;
            ++g_polish_ok;
            return 1;
        }
        for (int k = 0; k < m; ++k) w->yall[w->W[k]] = w->y[k];
        for (int k = 0; k < m; ++k) if (w->y[k] < -1e-10 * ymax) { w->inW[w->W[k]] = 0; w->yall[w->W[k]] = 0; }
        for (int i = 0; i < nc; ++i) if (!w->inW[i] && w->r[i] > 1e-12 * (fabs(w->h[i]) > 1.0 ? fabs(w->h[i]) : 1.0)) {
            int was = 0; for (int k = 0; k < m; ++k) if (w->W[k] == i) was = 1;
            if (!was) { w->inW[i] = 1; w->yall[i] = 0; }
        }
    }
    return 0;
}

/* returns status; w->z holds the scaled minimiser */
static int solve_dense(const form_t *f, const double *xk, const double *ref, double tol, int max_iter,
                       work_t *w, int *iters) {
    const int nv = f->nw, nc = f->nc, nx = f->nx;
    const double *Gs = f->Gs, *Hs = f->Hs;
    *iters = 0;
    if (f->always_infeasible) return TMPC_STATUS_INFEASIBLE;
    for (int r = 0; r < f->npar; ++r) {
        double v = f->gp0[r];
        for (int c = 0; c < nx; ++c) v += f->Ep[r * nx + c] * xk[c];
        if (v < -1e-9 * (1.0 + fabs(f->gp0[r]))) return TMPC_STATUS_INFEASIBLE;
    }
    for (int i = 0; i < nv; ++i) {
        double v = 0;
        for (int c = 0; c < nx; ++c) v += f->F1s[i * nx + c] * xk[c] + f->F2s[i * nx + c] * ref[c];
        w->q[i] = v;
    }
    for (int r = 0; r < nc; ++r) {
        double v = f->g0s[r];
        for (int c = 0; c < nx; ++c) v += f->Es[r * nx + c] * xk[c];
        w->h[r] = v;
    }
    /* unconstrained minimiser; done if feasible */
    for (int i = 0; i < nv; ++i) { double v = 0; for (int j = 0; j < nv; ++j) v -= f->Hinv[i * nv + j] * w->q[j]; w->z[i] = v; }
    double smin = INFINITY, qn = 1.0, hn = 1.0;
    for (int r = 0; r < nc; ++r) {
        double v = w->h[r];
        for (int j = 0; j < nv; ++j) v -= Gs[(size_t)r * nv + j] * w->z[j];
        w->s[r] = v; if (v < smin) smin = v;
        if (fabs(w->h[r]) > hn) hn = fabs(w->h[r]);
    }
    for (int i = 0; i < nv; ++i) if (fabs(w->q[i]) > qn) qn = fabs(w->q[i]);
    if (smin >= 0) { for (int r = 0; r < nc; ++r) w->lam[r] = 0; return TMPC_STATUS_OPTIMAL; }
    {
        double viol = -smin, fl = 0.1 * (viol > 1.0 ? viol : 1.0);
        /* Starting multipliers: the QP with row r alone has the multiplier viol_r / (g_r Hs^-1 g_r') at its minimiser; the
         * largest of them over the violated rows is the scale the multipliers have to reach.  lambda_0 is its fourth root,
         * between 1 and 1e3 (round 3: 1 for every problem; 3 - 7 % fewer iterations, a shorter upper tail).
         * Experiment knobs (developer only): ORACLE_INIT_FL scales the slack floor, ORACLE_INIT_LAM sets lambda_0 by hand,
         * ORACLE_INIT_MU > 0 uses the centred start lambda_i = mu / s_i instead */
        const char *e1 = getenv("ORACLE_INIT_FL"), *e2 = getenv("ORACLE_INIT_LAM"), *e3 = getenv("ORACLE_INIT_MU");
        const double cfl = e1 ? atof(e1) : 1.0, mu0 = e3 ? atof(e3) : 0.0;
        double l1 = 0.0;
        for (int r = 0; r < nc; ++r) {
            if (w->s[r] >= 0) continue;
            double c = 0;
            for (int i = 0; i < nv; ++i) {
                double t = 0;
                for (int j = 0; j < nv; ++j) t += f->Hinv[i * nv + j] * Gs[(size_t)r * nv + j];
                c += t * Gs[(size_t)r * nv + i];
            }
            if (c > 0 && -w->s[r] / c > l1) l1 = -w->s[r] / c;
        }
        double l0 = sqrt(sqrt(l1));
        l0 = l0 < 1.0 ? 1.0 : (l0 > 1e3 ? 1e3 : l0);
        if (e2) l0 = atof(e2);
        fl *= cfl;
        for (int r = 0; r < nc; ++r) { if (w->s[r] < fl) w->s[r] = fl; w->lam[r] = mu0 > 0 ? mu0 / w->s[r] : l0; }
    }
    double try_tol = tol;
    int status = TMPC_STATUS_MAX_ITER;
    for (int it = 0; it < max_iter; ++it) {
        *iters = it;
        /* residuals */
        double rdn = 0, rpn = 0, gap = 0, obj = 0, lmax = 0;
        for (int r = 0; r < nc; ++r) {
            double gz = 0;
            for (int j = 0; j < nv; ++j) gz += Gs[(size_t)r * nv + j] * w->z[j];
            w->gz[r] = gz;
            w->rp[r] = gz + w->s[r] - w->h[r];
            if (fabs(w->rp[r]) > rpn) rpn = fabs(w->rp[r]);
            gap += w->s[r] * w->lam[r];
            if (w->lam[r] > lmax) lmax = w->lam[r];
        }
        for (int i = 0; i < nv; ++i) {
            double hz = 0;
            for (int j = 0; j < nv; ++j) hz += Hs[i * nv + j] * w->z[j];
            obj += w->z[i] * (0.5 * hz + w->q[i]);
            w->tmp[i] = hz + w->q[i];                    /* gradient of the cost */
        }
        for (int i = 0; i < nv; ++i) {
            double v = w->tmp[i];
            for (int r = 0; r < nc; ++r) v += Gs[(size_t)r * nv + i] * w->lam[r];
            if (fabs(v) > rdn) rdn = fabs(v);
        }
        double mu = gap / nc;
        if (getenv("ORACLE_DEBUG")) fprintf(stderr, "it %d mu %.3e rd %.3e rp %.3e gap %.3e obj %.3e\n", it, mu, rdn / qn, rpn / hn, gap, obj);
        if (!(mu == mu) || !(rdn == rdn)) { status = TMPC_STATUS_NUMERICAL; break; }
        /* The interior-point phase only has to identify the active set: stationarity is
         * restored exactly by polish(), so r_d gets a looser threshold than r_p and the gap. */
        const double objs = fabs(obj) > 1.0 ? fabs(obj) : 1.0;
        if (rdn <= 1e3 * try_tol * qn && rpn <= try_tol * hn && gap <= try_tol * objs) {
            if (polish(f, w)) { status = TMPC_STATUS_OPTIMAL; break; }
            if (try_tol <= 1e-12) {
                status = (rdn <= 1e-9 * qn) ? TMPC_STATUS_OPTIMAL : TMPC_STATUS_MAX_ITER;
                break;
            }
            try_tol *= 1e-2;
        }
        if (gap <= 1e-15 * objs) {
            /* nothing left to gain from further iterations (the dual residual stalls on the ill-conditioned systems of such
             * a small mu and keeps the hand-over test above from passing): the refinement gets this iterate as it is */
            status = polish(f, w) ? TMPC_STATUS_OPTIMAL : TMPC_STATUS_MAX_ITER;
            break;
        }
        /* Farkas-type infeasibility test: lam blows up while G'lam -> 0 and h'lam < 0 */
        if (lmax > 1e10) {
            double hl = 0, gn = 0;
            for (int r = 0; r < nc; ++r) hl += w->h[r] * w->lam[r];
            for (int i = 0; i < nv; ++i) {
                double v = 0;
                for (int r = 0; r < nc; ++r) v += Gs[(size_t)r * nv + i] * w->lam[r];
                if (fabs(v) > gn) gn = fabs(v);
            }
            if (hl < 0 && gn <= 1e-6 * lmax) { status = TMPC_STATUS_INFEASIBLE; break; }
        }
        /* M = Hs + G' D G */
        for (int r = 0; r < nc; ++r) w->d[r] = w->lam[r] / w->s[r];
        memcpy(w->M, Hs, sizeof(double) * nv * nv);
        for (int r = 0; r < nc; ++r) {
            const double *g = Gs + (size_t)r * nv; const double d = w->d[r];
            for (int i = 0; i < nv; ++i) {
                double t = d * g[i];
                for (int j = 0; j <= i; ++j) w->M[i * nv + j] += t * g[j];
            }
        }
        memcpy(w->L, w->M, sizeof(double) * nv * nv);
        if (chol(w->L, nv)) {
            double tr = 0;
            for (int i = 0; i < nv; ++i) tr += w->M[i * nv + i];
            memcpy(w->L, w->M, sizeof(double) * nv * nv);
            for (int i = 0; i < nv; ++i) w->L[i * nv + i] += 1e-13 * tr;
            if (chol(w->L, nv)) { status = TMPC_STATUS_NUMERICAL; break; }
        }
        /* affine direction: M dz = -(Hs z + q) - G'(d .* rp)   (the lam terms cancel) */
        for (int i = 0; i < nv; ++i) {
            double v = -w->tmp[i];
            for (int r = 0; r < nc; ++r) v -= Gs[(size_t)r * nv + i] * (w->d[r] * w->rp[r]);
            w->rhs[i] = v; w->dz[i] = v;
        }
        chol_solve(w->L, nv, w->dz);
        double sl = 0, sdl = 0, dsdl = 0;
        for (int r = 0; r < nc; ++r) {
            double gdz = 0;
            for (int j = 0; j < nv; ++j) gdz += Gs[(size_t)r * nv + j] * w->dz[j];
            w->dsa[r] = -w->rp[r] - gdz;
            w->dla[r] = -w->lam[r] - w->d[r] * w->dsa[r];
        }
        double aaff = max_step(w->s, w->dsa, w->lam, w->dla, nc, 1.0);
        for (int r = 0; r < nc; ++r) {
            sl += (w->s[r] + aaff * w->dsa[r]) * (w->lam[r] + aaff * w->dla[r]);
        }
        (void)sdl; (void)dsdl;
        double sigma = sl / nc / mu; sigma = sigma * sigma * sigma; if (sigma > 1.0) sigma = 1.0;
        /* corrector: rhs += G' ((dsa .* dla - sigma mu) ./ s) */
        for (int i = 0; i < nv; ++i) {
            double v = w->rhs[i];
            for (int r = 0; r < nc; ++r) v += Gs[(size_t)r * nv + i] * ((w->dsa[r] * w->dla[r] - sigma * mu) / w->s[r]);
            w->dz[i] = v;
        }
        chol_solve(w->L, nv, w->dz);
        for (int r = 0; r < nc; ++r) {
            double gdz = 0;
            for (int j = 0; j < nv; ++j) gdz += Gs[(size_t)r * nv + j] * w->dz[j];
            w->ds[r] = -w->rp[r] - gdz;
            double rc = w->s[r] * w->lam[r] + w->dsa[r] * w->dla[r] - sigma * mu;
            w->dl[r] = -(rc + w->lam[r] * w->ds[r]) / w->s[r];
        }
        double om = (1.0 - aaff) * (1.0 - aaff);
        if (om < 1e-4) om = 1e-4;
        if (om > 1e-2) om = 1e-2;
        double a = max_step(w->s, w->ds, w->lam, w->dl, nc, 1.0 - om);
        for (int i = 0; i < nv; ++i) w->z[i] += a * w->dz[i];
        for (int r = 0; r < nc; ++r) { w->s[r] += a * w->ds[r]; w->lam[r] += a * w->dl[r]; }
        *iters = it + 1;
    }
    /* Iteration cap (or a stalled gap): the last iterate is returned under TMPC_STATUS_MAX_ITER, as the reference uses what an
     * inaccurate solve leaves in its variables (TubeTrackingMPC.py:185-192).  INFEASIBLE is only ever declared with the
     * Farkas-type certificate above. */
    return status;
}

/* ------------------------------------------------------------------ public API */
int oracle_create(const tmpc_problem *p, oracle_handle **out) {
    if (!p || !out || p->nx <= 0 || p->nu <= 0 || p->N <= 0) { snprintf(g_err, sizeof g_err, "invalid problem"); return TMPC_E_INVALID; }
    oracle_handle *h = (oracle_handle *)calloc(1, sizeof(*h));
    h->nx = p->nx; h->nu = p->nu; h->N = p->N;
    h->nvariants = p->extended ? 2 : 1;
    h->tol = p->tol > 0 ? p->tol : 1e-7;
    h->max_iter = p->max_iter > 0 ? p->max_iter : 60;
    for (int v = 0; v < h->nvariants; ++v) {
        int rc = build_form(p, v, &h->f[v]);
        if (rc) {
            snprintf(g_err, sizeof g_err, "oracle: building variant %d failed (code %d)", v, rc);
            for (int k = 0; k <= v; ++k) free_form(&h->f[k]);
            free(h);
            return TMPC_E_INVALID;
        }
    }
    *out = h;
    return TMPC_OK;
}

void oracle_destroy(oracle_handle *h) {
    if (!h) return;
    for (int v = 0; v < h->nvariants; ++v) free_form(&h->f[v]);
    free(h);
}

const char *oracle_last_error(void) { return g_err; }

int oracle_get_dims(const oracle_handle *h, int variant, int32_t *nv, int32_t *nc, int32_t *npar) {
    if (!h || variant < 0 || variant >= h->nvariants) return TMPC_E_INVALID;
    if (nv) *nv = h->f[variant].nw;
    if (nc) *nc = h->f[variant].nc;
    if (npar) *npar = h->f[variant].npar;
    return TMPC_OK;
}

int oracle_solve_batch(oracle_handle *h, int64_t B, const double *x_k, const double *ref, const uint8_t *variant,
                       double *u_nom, double *x_nom0, double *xu_ss, double *x_nom,
                       int32_t *status, int32_t *iters, int nthreads) {
    if (!h || B < 0) return TMPC_E_INVALID;
    const int nx = h->nx, nu = h->nu, N = h->N;
    int bad = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel
    {
        work_t *w[2] = {0, 0};
        double *v = dalloc(h->f[0].nvar + (h->nvariants > 1 ? h->f[1].nvar : 0) + 8);
        for (int k = 0; k < h->nvariants; ++k) w[k] = work_alloc(h->f[k].nw, h->f[k].nc);
#pragma omp for schedule(dynamic, 16)
        for (int64_t b = 0; b < B; ++b) {
            int var = variant ? variant[b] : 0;
            if (var < 0 || var >= h->nvariants) { bad = 1; continue; }
            const form_t *f = &h->f[var];
            const double *xk = x_k + b * nx, *rf = ref + b * nx;
            int it = 0;
            int st = solve_dense(f, xk, rf, h->tol, h->max_iter, w[var], &it);
            if (status) status[b] = st;
            if (iters) iters[b] = it;
            if (st >= TMPC_STATUS_INFEASIBLE) {
                for (int i = 0; i < N * nu; ++i) u_nom[b * N * nu + i] = NAN;
                if (x_nom0) for (int i = 0; i < nx; ++i) x_nom0[b * nx + i] = NAN;
                if (xu_ss) for (int i = 0; i < nx + nu; ++i) xu_ss[b * (nx + nu) + i] = NAN;
                if (x_nom) for (int i = 0; i < (N + 1) * nx; ++i) x_nom[b * (N + 1) * nx + i] = NAN;
                continue;
            }
            /* v = V0 x_k + Zb (Dv .* z) */
            for (int i = 0; i < f->nvar; ++i) {
                double a = 0;
                for (int c = 0; c < nx; ++c) a += f->V0[i * nx + c] * xk[c];
                for (int j = 0; j < f->nw; ++j) a += f->Zb[i * f->nw + j] * (f->Dv[j] * w[var]->z[j]);
                v[i] = a;
            }
            const int ou = nx * (N + 1), oxb = ou + nu * N;
            memcpy(u_nom + b * N * nu, v + ou, sizeof(double) * N * nu);
            if (x_nom0) memcpy(x_nom0 + b * nx, v, sizeof(double) * nx);
            if (xu_ss) memcpy(xu_ss + b * (nx + nu), v + oxb, sizeof(double) * (nx + nu));
            if (x_nom) memcpy(x_nom + b * (N + 1) * nx, v, sizeof(double) * (N + 1) * nx);
        }
        for (int k = 0; k < h->nvariants; ++k) work_free(w[k]);
        free(v);
    }
    return bad ? TMPC_E_INVALID : TMPC_OK;
}

/* exports the reduced, unscaled form for the tests (any pointer may be NULL) */
int oracle_get_reduced(const oracle_handle *h, int variant, double *H, double *F1, double *F2, double *G, double *g0, double *E) {
    if (!h || variant < 0 || variant >= h->nvariants) return TMPC_E_INVALID;
    const form_t *f = &h->f[variant];
    if (H) memcpy(H, f->H, sizeof(double) * f->nw * f->nw);
    if (F1) memcpy(F1, f->F1, sizeof(double) * f->nw * f->nx);
    if (F2) memcpy(F2, f->F2, sizeof(double) * f->nw * f->nx);
    if (G) memcpy(G, f->G, sizeof(double) * f->nc * f->nw);
    if (g0) memcpy(g0, f->g0, sizeof(double) * f->nc);
    if (E) memcpy(E, f->E, sizeof(double) * f->nc * f->nx);
    return TMPC_OK;
}
