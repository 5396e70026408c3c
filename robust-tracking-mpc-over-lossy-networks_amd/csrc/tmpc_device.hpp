// Device-side view of one condensed, scaled QP variant and the launch interface
// between tmpc_api.cpp (host) and tmpc_kernels.hip (device).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/tmpc.h"

namespace tmpc {

// All pointers are device pointers owned by the handle.  NVP = padded variable count,
// NCP = padded row count of the kernel instantiation chosen for this variant.
struct DeviceQP {
    int nx, nu, N;
    int nv, nc, npar, nth;
    int off_theta, off_x0, off_aux;
    int max_iter, always_infeasible;
    double tol;
    const double *Gt;     // [NVP][NCP]  scaled G, transposed, zero padded
    const double *Hs;     // [NVP][NVP]  scaled Hessian, identity on the padding
    const double *Hinv;   // [NVP][NVP]
    const double *F1s;    // [nv][nx]
    const double *F2s;    // [nv][nx]
    const double *g0s;    // [nc]
    const double *Es;     // [nc][nx]
    const double *gp0;    // [npar]
    const double *Ep;     // [npar][nx]
    const double *Dv;     // [nv]
    const double *Mth;    // [nx+nu][nth]
    const double *A;      // [nx][nx]
    const double *B;      // [nx][nu]
    long long *dbg;       // diagnostic builds only (TMPC_STAMPS); nullptr otherwise
};

// Chooses the smallest compiled (NVP, RPL) that covers (nv, nc); false if none does.
bool pick_config(int nv, int nc, int *nvp, int *rpl);
size_t lds_bytes(int nvp, int rpl);

hipError_t launch_solve(const DeviceQP &qp, int nvp, int rpl, int variant_id, int64_t B, const double *x_k,
                        const double *ref, const uint8_t *variant, double *u_nom, double *x_nom0, double *xu_ss,
                        double *x_nom, int32_t *status, int32_t *iters, int n_cu, hipStream_t stream);

}  // namespace tmpc
