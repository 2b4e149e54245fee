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


// ------------------------------------------------------------------------------------------------------------
// Whole half-steps of the generic NH-Verlet path as single launches (the analytic-adjoint / graph-replay path of
// SchNet and other non-fusable interactions): what NHVerlet.integrate and _analytic_nhc_adjoint (sovlers.py) spend
// ~30 + ~45 tiny tensor ops per MD step on.  One workgroup per replica; the step / frame index lives on the device
// (idx[0], int64: a captured HIP graph advances it itself), dt = t[k+1] - t[k] is read here.
__device__ __forceinline__ float nhc_bath(const float* pr, const float* Q, float T, float n_dof, float ke, int C, int k) {
    if (k == 0) return 2.f * (ke - T * n_dof * 0.5f) - pr[0] * pr[1] / Q[1];
    if (k == C - 1) return pr[C - 2] * pr[C - 2] / Q[C - 2] - T;
    return (pr[k - 1] * pr[k - 1] / Q[k - 1] - T) - pr[k + 1] * pr[k] / Q[k + 1];
}
__device__ __forceinline__ float nhc_bath_vjp(const float* pr, const float* lr, const float* Q, float slv, int C, int k) {
    if (k == 0) return -slv / Q[0] - lr[0] * pr[1] / Q[1] + 2.f * pr[0] * lr[1] / Q[0];
    if (k == C - 1) return -lr[C - 2] * pr[C - 2] / Q[C - 1];
    return -lr[k - 1] * pr[k - 1] / Q[k] - lr[k] * pr[k + 1] / Q[k + 1] + 2.f * pr[k] * lr[k + 1] / Q[k];
}

// A replica's 3n elements are cut into NHV_CHUNK-element chunks, one workgroup each (grid = chunks x replicas): at
// 4 096 beads a single 256-thread workgroup per replica left the launch latency-bound (~16 us; 12 workgroups ~5 us).
// The kinetic-energy / <lam_v, v> sums cross workgroups through `scratch` (per-chunk partials + one ticket per
// replica): the workgroup that draws the last ticket adds the partials IN CHUNK ORDER (so the result does not depend on
// which workgroup finished last) and does the chain update; it also hands the ticket back at zero for the next launch.
constexpr int NHV_CHUNK = 1024;
__host__ __device__ inline int nhv_chunks(int n) { return (3 * n + NHV_CHUNK - 1) / NHV_CHUNK; }

// `idx` (nullable): the launch also moves the device-side step / frame counter to `new_idx` -- by the workgroup that draws
// the last ticket of the WHOLE grid, i.e. after every workgroup has read the old value (each reads it before its loop).
template <int NV>
__device__ __forceinline__ bool replica_sum(float (&val)[NV], float* red, float* scratch, int R, int r, long long* idx = nullptr,
                                            long long new_idx = 0) {
    const int nb = gridDim.x;
#pragma unroll
    for (int i = 0; i < NV; ++i) val[i] = block_sum(val[i], red);
    __shared__ int last;
    float* part = scratch + (size_t)r * nb * 2;
    unsigned* tickets = reinterpret_cast<unsigned*>(scratch + (size_t)R * nb * 2);
    if (threadIdx.x == 0) {
        int mine = 1;
        if (nb > 1) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                __hip_atomic_store(part + blockIdx.x * 2 + i, val[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence();
            const unsigned tk = atomicAdd(tickets + r, 1u);
            mine = tk == (unsigned)(nb - 1);
            if (mine) __hip_atomic_store(tickets + r, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        last = mine;
        if (idx) {
            const unsigned all = gridDim.x * gridDim.y;
            if (all == 1) idx[0] = new_idx;
            else if (atomicAdd(tickets + R, 1u) == all - 1) {
                __hip_atomic_store(tickets + R, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                idx[0] = new_idx;
            }
        }
    }
    if (nb == 1 && !idx) return true;
    __syncthreads();
    if (!last) return false;
    if (nb == 1) return true;
    __threadfence();
    // (round 6) the partials are fetched by nb threads at once and added from LDS in chunk order -- the same sum, bit for bit,
    // as the former loop of nb dependent device-scope loads per thread (12 x ~0.7 us at 4 096 beads: most of these launches)
    constexpr int STAGE = 256;
    __shared__ float stage[NV][STAGE];
    if (nb <= STAGE) {
        if ((int)threadIdx.x < nb) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                stage[i][threadIdx.x] = __hip_atomic_load(part + threadIdx.x * 2 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float s = 0.f;
            for (int b = 0; b < nb; ++b) s += stage[i][b];
            val[i] = s;
        }
        return true;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = 0.f;
        for (int b = 0; b < nb; ++b) s += __hip_atomic_load(part + b * 2 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        val[i] = s;
    }
    return true;
}

#define NHV_RANGE                                                                                   \
    const int r = blockIdx.y;                                                                       \
    const size_t o = (size_t)r * n * 3;                                                             \
    const int e0 = blockIdx.x * NHV_CHUNK, e1 = min(e0 + NHV_CHUNK, 3 * n)

// forward, first half (sovlers.py:111-118): rhs at (v, q, pv) with the cached force ->
//   dv_h = 1/2 a dt, dp_h = 1/2 b dt, qn = q + (v + dv_h) dt
__global__ __launch_bounds__(NHC_BLOCK) void nhv_kick_kernel(
    const float* __restrict__ v, const float* __restrict__ q, const float* __restrict__ pv, const float* __restrict__ f,
    const float* __restrict__ mass, const float* __restrict__ Q, const float* __restrict__ Tp, float n_dof,
    const float* __restrict__ t, const long long* __restrict__ idx, int R, int n, int C, float* __restrict__ dv_h,
    float* __restrict__ dp_h, float* __restrict__ qn, float* __restrict__ scratch) {
    __shared__ float red[32];
    __shared__ float ps[MDG_MAX_CHAINS];
    NHV_RANGE;
    const long long k = idx[0];
    const float dt = t[k + 1] - t[k], T = Tp[0];
    const float* pr = pv + (size_t)r * C;
    const float pv0 = pr[0], q0 = Q[0];
    float s[1] = {0.f};
    for (int e = e0 + threadIdx.x; e < e1; e += NHC_BLOCK) {
        const float m = mass[(size_t)r * n + e / 3], ve = v[o + e], p = ve * m;
        s[0] += p * p / m;
        const float a = (f[o + e] - pv0 * p / q0) / m;
        const float h = 1.f / 2.f * a * dt;
        dv_h[o + e] = h;
        qn[o + e] = q[o + e] + (ve + h) * dt;
    }
    if (!replica_sum(s, red, scratch, R, r)) return;
    const float ke = 0.5f * s[0];
    if (threadIdx.x < C) ps[threadIdx.x] = pr[threadIdx.x];
    __syncthreads();
    if (threadIdx.x < C) dp_h[(size_t)r * C + threadIdx.x] = 1.f / 2.f * nhc_bath(ps, Q, T, n_dof, ke, C, threadIdx.x) * dt;
}

// forward, second half (sovlers.py:121-125): rhs at (v + dv_h, qn, pv + dp_h) with the new force ->
//   v += dv_h + 1/2 a1 dt, pv += dp_h + 1/2 b1 dt, q = qn, f = fn; frame k+1 stored
// (pv is written by the last workgroup only, after every workgroup of the replica has read it)
__global__ __launch_bounds__(NHC_BLOCK) void nhv_finish_kernel(
    float* __restrict__ v, float* __restrict__ q, float* __restrict__ pv, float* __restrict__ f,
    const float* __restrict__ dv_h, const float* __restrict__ dp_h, const float* __restrict__ qn,
    const float* __restrict__ fn, const float* __restrict__ mass, const float* __restrict__ Q,
    const float* __restrict__ Tp, float n_dof, const float* __restrict__ t, long long* idx, int advance, int R, int n,
    int C, float* __restrict__ out_v, float* __restrict__ out_q, float* __restrict__ out_pv, float* __restrict__ scratch) {
    __shared__ float red[32];
    __shared__ float ps[MDG_MAX_CHAINS];
    NHV_RANGE;
    const long long k = idx[0];
    const float dt = t[k + 1] - t[k], T = Tp[0];
    float* pr = pv + (size_t)r * C;
    if (threadIdx.x < C) ps[threadIdx.x] = pr[threadIdx.x] + dp_h[(size_t)r * C + threadIdx.x];
    __syncthreads();
    const float pv0 = ps[0], q0 = Q[0];
    const size_t fo = (size_t)(k + 1) * R * n * 3 + o;              // frames are time-major: [T, R*n, 3]
    float s[1] = {0.f};
    for (int e = e0 + threadIdx.x; e < e1; e += NHC_BLOCK) {
        const float m = mass[(size_t)r * n + e / 3], h = dv_h[o + e], ve = v[o + e] + h, p = ve * m;
        s[0] += p * p / m;
        const float a1 = (fn[o + e] - pv0 * p / q0) / m;
        const float vn = v[o + e] + (h + 1.f / 2.f * a1 * dt);
        const float qe = qn[o + e];
        v[o + e] = vn; q[o + e] = qe; f[o + e] = fn[o + e];
        out_v[fo + e] = vn; out_q[fo + e] = qe;
    }
    if (!replica_sum(s, red, scratch, R, r, advance ? idx : nullptr, k + 1)) return;
    const float ke = 0.5f * s[0];
    if (threadIdx.x < C) {
        const float pn = pr[threadIdx.x] + (dp_h[(size_t)r * C + threadIdx.x] + 1.f / 2.f * nhc_bath(ps, Q, T, n_dof, ke, C, threadIdx.x) * dt);
        pr[threadIdx.x] = pn;
        out_pv[((size_t)(k + 1) * R + r) * C + threadIdx.x] = pn;
    }
}

// adjoint, before the first evaluation of interval i = idx[0]: frame i of the saved trajectory as contiguous
// (v, q, pv) and the adjoint direction w = lam_v / m
__global__ __launch_bounds__(NHC_BLOCK) void nhv_adj_pre_kernel(
    const float* __restrict__ v_t, const float* __restrict__ q_t, const float* __restrict__ pv_t,
    const float* __restrict__ lv, const float* __restrict__ mass, const long long* __restrict__ idx, int R, int n, int C,
    float* __restrict__ v, float* __restrict__ q, float* __restrict__ pv, float* __restrict__ w) {
    NHV_RANGE;
    const size_t fo = (size_t)idx[0] * R * n * 3 + o;
    for (int e = e0 + threadIdx.x; e < e1; e += NHC_BLOCK) {
        v[o + e] = v_t[fo + e]; q[o + e] = q_t[fo + e];
        w[o + e] = lv[o + e] / mass[(size_t)r * n + e / 3];
    }
    if (blockIdx.x == 0 && threadIdx.x < C) pv[(size_t)r * C + threadIdx.x] = pv_t[((size_t)idx[0] * R + r) * C + threadIdx.x];
}

// adjoint, after the first evaluation (F, dwF_dq at frame i): midpoint state and half-step adjoint
// (sovlers.py:132-145): vh = v - a hh, qm = q + vh h, pm = pv - b hh, lam_h = lam + G0 hh, w_h = lam_h_v / m
__global__ __launch_bounds__(NHC_BLOCK) void nhv_adj_mid_kernel(
    const float* __restrict__ v, const float* __restrict__ q, const float* __restrict__ pv, const float* __restrict__ lv,
    const float* __restrict__ lq, const float* __restrict__ lp, const float* __restrict__ f, const float* __restrict__ dwf,
    const float* __restrict__ mass, const float* __restrict__ Q, const float* __restrict__ Tp, float n_dof,
    const float* __restrict__ t, const long long* __restrict__ idx, int R, int n, int C, float* __restrict__ vh,
    float* __restrict__ qm, float* __restrict__ pm, float* __restrict__ lvh, float* __restrict__ lqh,
    float* __restrict__ lph, float* __restrict__ wh, float* __restrict__ scratch) {
    __shared__ float red[32];
    __shared__ float ps[MDG_MAX_CHAINS], ls[MDG_MAX_CHAINS];
    NHV_RANGE;
    const long long i = idx[0];
    const float h = t[i] - t[i - 1], hh = 0.5f * h, T = Tp[0];
    if (threadIdx.x < C) { ps[threadIdx.x] = pv[(size_t)r * C + threadIdx.x]; ls[threadIdx.x] = lp[(size_t)r * C + threadIdx.x]; }
    __syncthreads();
    const float pv0 = ps[0], lp0 = ls[0], q0 = Q[0];
    float s[2] = {0.f, 0.f};
    for (int e = e0 + threadIdx.x; e < e1; e += NHC_BLOCK) {
        const float m = mass[(size_t)r * n + e / 3], ve = v[o + e], p = ve * m, le = lv[o + e];
        s[0] += p * p / m; s[1] += le * ve;
        const float a = (f[o + e] - pv0 * p / q0) / m;
        const float Gv = -(pv0 / q0) * le + lq[o + e] + 2.f * m * ve * lp0;
        const float vhe = ve - a * hh;                                          // :132
        vh[o + e] = vhe;
        qm[o + e] = q[o + e] + vhe * h;                                         // :138 (forward-time sign)
        const float lvhe = le + Gv * hh;                                        // :141
        lvh[o + e] = lvhe;
        lqh[o + e] = lq[o + e] + dwf[o + e] * hh;                               // :142
        wh[o + e] = lvhe / m;
    }
    if (!replica_sum(s, red, scratch, R, r)) return;
    const float ke = 0.5f * s[0], slv = s[1];
    if (threadIdx.x < C) {
        pm[(size_t)r * C + threadIdx.x] = ps[threadIdx.x] - nhc_bath(ps, Q, T, n_dof, ke, C, threadIdx.x) * hh;      // :135
        lph[(size_t)r * C + threadIdx.x] = ls[threadIdx.x] + nhc_bath_vjp(ps, ls, Q, slv, C, threadIdx.x) * hh;     // :143
    }
}

// adjoint, after the midpoint evaluation (dwF_dq at the midpoint): lam += G1 h + dL/dy_{i-1}   (sovlers.py:156-158, :286)
__global__ __launch_bounds__(NHC_BLOCK) void nhv_adj_end_kernel(
    const float* __restrict__ vh, const float* __restrict__ pm, const float* __restrict__ lvh, const float* __restrict__ lqh,
    const float* __restrict__ lph, const float* __restrict__ dwf, const float* __restrict__ mass,
    const float* __restrict__ Q, const float* __restrict__ t, long long* idx, int advance,
    const float* __restrict__ g_v, const float* __restrict__ g_q, const float* __restrict__ g_pv, int R, int n, int C,
    float* __restrict__ lv, float* __restrict__ lq, float* __restrict__ lp, float* __restrict__ scratch) {
    __shared__ float red[32];
    __shared__ float ps[MDG_MAX_CHAINS], ls[MDG_MAX_CHAINS];
    NHV_RANGE;
    const long long i = idx[0];
    const float h = t[i] - t[i - 1];
    const size_t go = (size_t)(i - 1) * R * n * 3 + o;
    if (threadIdx.x < C) { ps[threadIdx.x] = pm[(size_t)r * C + threadIdx.x]; ls[threadIdx.x] = lph[(size_t)r * C + threadIdx.x]; }
    __syncthreads();
    const float pv0 = ps[0], lp0 = ls[0], q0 = Q[0];
    float s[1] = {0.f};
    for (int e = e0 + threadIdx.x; e < e1; e += NHC_BLOCK) {
        const float m = mass[(size_t)r * n + e / 3], ve = vh[o + e], le = lvh[o + e];
        s[0] += le * ve;
        const float Gv = -(pv0 / q0) * le + lqh[o + e] + 2.f * m * ve * lp0;
        lv[o + e] = lv[o + e] + Gv * h + g_v[go + e];
        lq[o + e] = lq[o + e] + dwf[o + e] * h + g_q[go + e];
    }
    if (!replica_sum(s, red, scratch, R, r, advance ? idx : nullptr, i - 1)) return;
    const float slv = s[0];
    if (threadIdx.x < C)
        lp[(size_t)r * C + threadIdx.x] = lp[(size_t)r * C + threadIdx.x] + nhc_bath_vjp(ps, ls, Q, slv, C, threadIdx.x) * h +
                                          g_pv[((size_t)(i - 1) * R + r) * C + threadIdx.x];
}
#undef NHV_RANGE

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

// floats of the cross-workgroup scratch the mdg_nhv_* launches of one (n_rep, n_atoms) share; ZERO it once
extern "C" int64_t mdg_nhv_scratch_floats(int n_rep, int n_atoms) {
    if (n_rep <= 0 || n_atoms <= 0) return 0;
    return (int64_t)n_rep * nhv_chunks(n_atoms) * 2 + n_rep + 1;
}

extern "C" int mdg_nhv_kick(const float* v, const float* q, const float* pv, const float* f, const float* mass, const float* Q,
                            const float* T, float n_dof, const float* t, const int64_t* idx, int n_rep, int n_atoms,
                            int n_chains, float* dv_h, float* dp_h, float* qn, float* scratch, void* stream) {
    MDG_CHECK_ARG(v && q && pv && f && mass && Q && T && t && idx && dv_h && dp_h && qn && scratch, "nhv_kick: null buffer");
    MDG_CHECK_ARG(n_rep > 0 && n_atoms > 0 && n_chains >= 2 && n_chains <= MDG_MAX_CHAINS, "nhv_kick: bad sizes");
    hipLaunchKernelGGL(nhv_kick_kernel, dim3(nhv_chunks(n_atoms), n_rep), dim3(NHC_BLOCK), 0, (hipStream_t)stream, v, q, pv, f,
                       mass, Q, T, n_dof, t, reinterpret_cast<const long long*>(idx), n_rep, n_atoms, n_chains, dv_h, dp_h, qn,
                       scratch);
    MDG_CHECK_LAUNCH("nhv_kick_kernel");
    return MDG_OK;
}

extern "C" int mdg_nhv_finish(float* v, float* q, float* pv, float* f, const float* dv_h, const float* dp_h, const float* qn,
                              const float* fn, const float* mass, const float* Q, const float* T, float n_dof,
                              const float* t, int64_t* idx, int advance, int n_rep, int n_atoms, int n_chains, float* out_v,
                              float* out_q, float* out_pv, float* scratch, void* stream) {
    MDG_CHECK_ARG(v && q && pv && f && dv_h && dp_h && qn && fn && mass && Q && T && t && idx && out_v && out_q && out_pv &&
                  scratch, "nhv_finish: null buffer");
    MDG_CHECK_ARG(n_rep > 0 && n_atoms > 0 && n_chains >= 2 && n_chains <= MDG_MAX_CHAINS, "nhv_finish: bad sizes");
    hipLaunchKernelGGL(nhv_finish_kernel, dim3(nhv_chunks(n_atoms), n_rep), dim3(NHC_BLOCK), 0, (hipStream_t)stream, v, q, pv,
                       f, dv_h, dp_h, qn, fn, mass, Q, T, n_dof, t, reinterpret_cast<long long*>(idx), advance, n_rep, n_atoms,
                       n_chains, out_v, out_q, out_pv, scratch);
    MDG_CHECK_LAUNCH("nhv_finish_kernel");
    return MDG_OK;
}

extern "C" int mdg_nhv_adj_pre(const float* v_t, const float* q_t, const float* pv_t, const float* lv, const float* mass,
                               const int64_t* idx, int n_rep, int n_atoms, int n_chains, float* v, float* q, float* pv,
                               float* w, void* stream) {
    MDG_CHECK_ARG(v_t && q_t && pv_t && lv && mass && idx && v && q && pv && w, "nhv_adj_pre: null buffer");
    MDG_CHECK_ARG(n_rep > 0 && n_atoms > 0, "nhv_adj_pre: bad sizes");
    hipLaunchKernelGGL(nhv_adj_pre_kernel, dim3(nhv_chunks(n_atoms), n_rep), dim3(NHC_BLOCK), 0, (hipStream_t)stream, v_t, q_t,
                       pv_t, lv, mass, reinterpret_cast<const long long*>(idx), n_rep, n_atoms, n_chains, v, q, pv, w);
    MDG_CHECK_LAUNCH("nhv_adj_pre_kernel");
    return MDG_OK;
}

extern "C" int mdg_nhv_adj_mid(const float* v, const float* q, const float* pv, const float* lv, const float* lq,
                               const float* lp, const float* f, const float* dwf, const float* mass, const float* Q,
                               const float* T, float n_dof, const float* t, const int64_t* idx, int n_rep, int n_atoms,
                               int n_chains, float* vh, float* qm, float* pm, float* lvh, float* lqh, float* lph, float* wh,
                               float* scratch, void* stream) {
    MDG_CHECK_ARG(v && q && pv && lv && lq && lp && f && dwf && mass && Q && T && t && idx && vh && qm && pm && lvh && lqh &&
                  lph && wh && scratch, "nhv_adj_mid: null buffer");
    MDG_CHECK_ARG(n_rep > 0 && n_atoms > 0 && n_chains >= 2 && n_chains <= MDG_MAX_CHAINS, "nhv_adj_mid: bad sizes");
    hipLaunchKernelGGL(nhv_adj_mid_kernel, dim3(nhv_chunks(n_atoms), n_rep), dim3(NHC_BLOCK), 0, (hipStream_t)stream, v, q, pv,
                       lv, lq, lp, f, dwf, mass, Q, T, n_dof, t, reinterpret_cast<const long long*>(idx), n_rep, n_atoms,
                       n_chains, vh, qm, pm, lvh, lqh, lph, wh, scratch);
    MDG_CHECK_LAUNCH("nhv_adj_mid_kernel");
    return MDG_OK;
}

extern "C" int mdg_nhv_adj_end(const float* vh, const float* pm, const float* lvh, const float* lqh, const float* lph,
                               const float* dwf, const float* mass, const float* Q, const float* t, int64_t* idx, int advance,
                               const float* g_v, const float* g_q, const float* g_pv, int n_rep, int n_atoms, int n_chains,
                               float* lv, float* lq, float* lp, float* scratch, void* stream) {
    MDG_CHECK_ARG(vh && pm && lvh && lqh && lph && dwf && mass && Q && t && idx && g_v && g_q && g_pv && lv && lq && lp &&
                  scratch, "nhv_adj_end: null buffer");
    MDG_CHECK_ARG(n_rep > 0 && n_atoms > 0 && n_chains >= 2 && n_chains <= MDG_MAX_CHAINS, "nhv_adj_end: bad sizes");
    hipLaunchKernelGGL(nhv_adj_end_kernel, dim3(nhv_chunks(n_atoms), n_rep), dim3(NHC_BLOCK), 0, (hipStream_t)stream, vh, pm,
                       lvh, lqh, lph, dwf, mass, Q, t, reinterpret_cast<long long*>(idx), advance, g_v, g_q, g_pv, n_rep, n_atoms,
                       n_chains, lv, lq, lp, scratch);
    MDG_CHECK_LAUNCH("nhv_adj_end_kernel");
    return MDG_OK;
}
