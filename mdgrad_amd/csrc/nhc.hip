// Nose-Hoover-chain right-hand side and its analytic vjp as single launches for the generic
// (non-fused) integrator path -- replaces the ~25 tiny elementwise/reduction ops of
// NoseHooverChain.forward after the force is known (torchmd/md.py:221-240) and the ~30 of
// the thermostat part of its vjp.  One workgroup per replica of a replica-stacked
// state ([R*n, 3] rows, chains [R, C]); block reductions in a fixed order (reproducible).
#include "common.hpp"

namespace {

constexpr int NHC_BLOCK = 256;

// a = (F - pv0 p / Q0) / m ,  p = m v ;  dpv = bath rhs (md.py:234-236) with KE = 1/2 sum p^2/m
__global__ __launch_bounds__(NHC_BLOCK) void nhc_rhs_kernel(
    const float* __restrict__ v, const float* __restrict__ f, const float* __restrict__ pv,
    const float* __restrict__ mass, const float* __restrict__ Q, const float* __restrict__ Tp, float n_dof, int n,
    int C, float* __restrict__ a, float* __restrict__ dpv) {
    __shared__ float red[32];
    const float T = Tp[0];            // thermostat temperature read from the device: a captured graph follows update_T
    const int r = blockIdx.x;
    const float* vr = v + (size_t)r * n * 3;
    const float* fr = f + (size_t)r * n * 3;
    const float* mr = mass + (size_t)r * n;
    const float* pr = pv + (size_t)r * C;
    const float pv0 = pr[0], q0 = Q[0];
    float part = 0.f;
    for (int e = threadIdx.x; e < 3 * n; e += NHC_BLOCK) {
        const float m = mr[e / 3], ve = vr[e], p = ve * m;
        part += p * p / m;
        a[(size_t)r * n * 3 + e] = (fr[e] - pv0 * p / q0) / m;
    }
    const float ke = 0.5f * block_sum(part, red);
    for (int k = threadIdx.x; k < C; k += NHC_BLOCK) {
        float d;
        if (k == 0) d = 2.f * (ke - T * n_dof * 0.5f) - pr[0] * pr[1] / Q[1];
        else if (k == C - 1) d = pr[C - 2] * pr[C - 2] / Q[C - 2] - T;
        else d = (pr[k - 1] * pr[k - 1] / Q[k - 1] - T) - pr[k + 1] * pr[k] / Q[k + 1];
        dpv[(size_t)r * C + k] = d;
    }
}

// Gv = -(pv0/Q0) lv + lq + 2 m v lp0 ;  Gp = lam^T d(bath rhs)/d pv + coupling  
__global__ __launch_bounds__(NHC_BLOCK) void nhc_vjp_kernel(
    const float* __restrict__ v, const float* __restrict__ pv, const float* __restrict__ lv,
    const float* __restrict__ lq, const float* __restrict__ lp, const float* __restrict__ mass,
    const float* __restrict__ Q, int n, int C, float* __restrict__ Gv, float* __restrict__ Gp) {
    __shared__ float red[32];
    const int r = blockIdx.x;
    const size_t o = (size_t)r * n * 3;
    const float* pr = pv + (size_t)r * C;
    const float* lr = lp + (size_t)r * C;
    const float pv0 = pr[0], lp0 = lr[0], q0 = Q[0];
    float part = 0.f;
    for (int e = threadIdx.x; e < 3 * n; e += NHC_BLOCK) {
        const float m = mass[(size_t)r * n + e / 3], ve = v[o + e], le = lv[o + e];
        part += le * ve;
        Gv[o + e] = -(pv0 / q0) * le + lq[o + e] + 2.f * m * ve * lp0;
    }
    const float slv = block_sum(part, red);
    for (int k = threadIdx.x; k < C; k += NHC_BLOCK) {
        float g;
        if (k == 0) g = -slv / Q[0] - lr[0] * pr[1] / Q[1] + 2.f * pr[0] * lr[1] / Q[0];
        else if (k == C - 1) g = -lr[C - 2] * pr[C - 2] / Q[C - 1];
        else g = -lr[k - 1] * pr[k - 1] / Q[k] - lr[k] * pr[k + 1] / Q[k + 1] + 2.f * pr[k] * lr[k + 1] / Q[k];
        Gp[(size_t)r * C + k] = g;
    }
}

}  // namespace

extern "C" int mdg_nhc_rhs(const float* v, const float* f, const float* pv, const float* mass, const float* Q,
                           const float* T, float n_dof, int n_rep, int n_atoms, int n_chains, float* a, float* dpv,
                           void* stream) {
    MDG_CHECK_ARG(v && f && pv && mass && Q && T && a && dpv, "nhc_rhs: null buffer");
    MDG_CHECK_ARG(n_rep > 0 && n_atoms > 0 && n_chains >= 2, "nhc_rhs: bad sizes R=%d n=%d C=%d", n_rep, n_atoms, n_chains);
    hipLaunchKernelGGL(nhc_rhs_kernel, dim3(n_rep), dim3(NHC_BLOCK), 0, (hipStream_t)stream, v, f, pv, mass, Q, T,
                       n_dof, n_atoms, n_chains, a, dpv);
    MDG_CHECK_LAUNCH("nhc_rhs_kernel");
    return MDG_OK;
}

extern "C" int mdg_nhc_vjp(const float* v, const float* pv, const float* lv, const float* lq, const float* lp,
                           const float* mass, const float* Q, int n_rep, int n_atoms, int n_chains, float* Gv,
                           float* Gp, void* stream) {
    MDG_CHECK_ARG(v && pv && lv && lq && lp && mass && Q && Gv && Gp, "nhc_vjp: null buffer");
    MDG_CHECK_ARG(n_rep > 0 && n_atoms > 0 && n_chains >= 2, "nhc_vjp: bad sizes R=%d n=%d C=%d", n_rep, n_atoms, n_chains);
    hipLaunchKernelGGL(nhc_vjp_kernel, dim3(n_rep), dim3(NHC_BLOCK), 0, (hipStream_t)stream, v, pv, lv, lq, lp, mass, Q,
                       n_atoms, n_chains, Gv, Gp);
    MDG_CHECK_LAUNCH("nhc_vjp_kernel");
    return MDG_OK;
}
