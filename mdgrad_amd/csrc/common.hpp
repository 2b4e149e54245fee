// Shared device code for libmdgrad_hip (gfx950 / CDNA4 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/mdgrad_hip.h"

#define MDG_WAVE 64

void mdg_set_error(const char* fmt, ...);

#define MDG_CHECK_ARG(cond, ...)                                                       \
    do { if (!(cond)) { mdg_set_error(__VA_ARGS__); return MDG_EINVAL; } } while (0)

// runtime calls on the launch path (stream-ordered copies / fills): fail loudly, with the call site
#define MDG_HIP(call)                                                                  \
    do { hipError_t e_ = (call);                                                       \
         if (e_ != hipSuccess) { mdg_set_error("%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
                                 return MDG_ELAUNCH; } } while (0)

#define MDG_CHECK_LAUNCH(name)                                                         \
    do { hipError_t e_ = hipGetLastError();                                            \
         if (e_ != hipSuccess) { mdg_set_error("%s: %s", name, hipGetErrorString(e_)); \
                                 return MDG_ELAUNCH; } } while (0)

// Fixed-point accumulation of float contributions (the table-gradient scatter of the tabulated pair model, csrc/traj_small.hip,
// traj_ring.hpp, traj_large.hip): ONE signed 64-bit word per value, integer atomics -- order-independent, so bitwise
// reproducible.  fx64(v) = round(v) exactly for |v| < 2^51 (two float roundings: the part above 2^20 and the rest).
// Range: a launch flags a single contribution at or above  fx64_limit(n) = 2^62 / n,  n = the largest number of
// contributions one word can receive, so no sum can leave int64 unflagged (two int32 planes wrapped silently once
// the adjoint had grown by ~2^10); the host re-runs a flagged launch with a coarser scale.
__device__ __forceinline__ unsigned long long fx64(float v) {
    const float hi = rintf(v * (1.f / 1048576.f));
    const int lo = (int)rintf(fmaf(hi, -1048576.f, v));
    return (unsigned long long)(((long long)(int)hi << 20) + (long long)lo);
}
__host__ __device__ inline float fx64_limit(double n_contrib) {
    const double lim = 4611686018427387904.0 / (n_contrib < 1.0 ? 1.0 : n_contrib);      // 2^62 / n
    return (float)(lim < 1125899906842624.0 ? lim : 1125899906842624.0);                 // (<= 2^50: fx64's own range)
}

// internal: the fine grids of the many-frame RDF kernels (csrc/rdf.hip), shared with the fused observable of the
// trajectory kernels (csrc/traj_small.hip).  Not part of the C ABI.
struct RdfFinePlan {
    float sc;                  // exp(coeff x^2) = exp2(-(sc x)^2)
    float reach, h;            // forward: fine integer histogram over [mu0 - reach, ...), bin width h
    long long nfine;           //          bins (0: does not fit)
    int reach_bins, ncell;     // backward: R of the derivative table, cells of the cubic-Hermite table (0: no table)
};
RdfFinePlan mdg_rdf_fine_plan(float spacing, float coeff, int nbins);
int mdg_rdf_fine_finish(const uint32_t* ghist, const RdfFinePlan& P, const float* mu, int nbins, float* raw, hipStream_t st);
int mdg_rdf_bwd_table(const float* mu, float coeff, int nbins, const float* g_raw, const RdfFinePlan& P, float4* tab,
                      hipStream_t st);
// ... the same cells, equally spaced in u = d^2 over the same distance range, holding (dL/dd)/d: a consumer that has d^2
// (the ring adjoint, csrc/traj_ring.hpp) needs neither the square root nor the division by d
int mdg_rdf_bwd_table_u(const float* mu, float coeff, int nbins, const float* g_raw, const RdfFinePlan& P, float4* tab,
                        hipStream_t st);
// grid of that table, derived from the (device-resident) centres: P.ncell cells over [xlo^2, xhi^2), xlo = mu0 - (R + 1) dmu,
// xhi = xlo + ncell dmu / 8  (R = P.reach_bins; the r-space grid of csrc/rdf.hip's fine_grid())
__device__ __forceinline__ void rdf_u_grid(const float* __restrict__ mu, int nbins, int R, float& ulo, float& hu, int& ncell) {
    const float mu0 = mu[0], dmu = (mu[nbins - 1] - mu0) / (float)(nbins - 1);
    ncell = (nbins - 1 + 2 * (R + 1)) * 8;
    const float xlo = mu0 - (float)(R + 1) * dmu, xhi = fmaf((float)ncell, dmu * 0.125f, xlo);
    ulo = xlo * xlo;
    hu = (xhi * xhi - ulo) / (float)ncell;
}

// ----------------------------------------------------------------------------- cell
// Minimum image exactly as topology.py:59-64: s = D . inv ; o = -(s > .5) + (s < -.5) ;
// D += o . h   (D = x_j - x_i).  Returns the packed image code (ox+1)+3(oy+1)+9(oz+1).
template <bool DIAG>
__device__ __forceinline__ int min_image(const MdgCell& c, float& dx, float& dy, float& dz) {
    float sx, sy, sz;
    if (DIAG) {
        sx = dx * c.inv[0]; sy = dy * c.inv[4]; sz = dz * c.inv[8];
    } else {
        sx = fmaf(dz, c.inv[6], fmaf(dy, c.inv[3], dx * c.inv[0]));
        sy = fmaf(dz, c.inv[7], fmaf(dy, c.inv[4], dx * c.inv[1]));
        sz = fmaf(dz, c.inv[8], fmaf(dy, c.inv[5], dx * c.inv[2]));
    }
    // -(s > .5) + (s < -.5)  ==  -clamp(rint(s), -1, 1): round-half-even maps the ties at
    // +-0.5 to 0 exactly like the strict comparisons; the clamp reproduces |s| >= 1.5.
    const float ox = -__builtin_amdgcn_fmed3f(rintf(sx), -1.f, 1.f);
    const float oy = -__builtin_amdgcn_fmed3f(rintf(sy), -1.f, 1.f);
    const float oz = -__builtin_amdgcn_fmed3f(rintf(sz), -1.f, 1.f);
    if (DIAG) {
        dx = fmaf(ox, c.h[0], dx); dy = fmaf(oy, c.h[4], dy); dz = fmaf(oz, c.h[8], dz);
    } else {
        dx += fmaf(oz, c.h[6], fmaf(oy, c.h[3], ox * c.h[0]));
        dy += fmaf(oz, c.h[7], fmaf(oy, c.h[4], ox * c.h[1]));
        dz += fmaf(oz, c.h[8], fmaf(oy, c.h[5], ox * c.h[2]));
    }
    return (int)(ox + 1.f) + 3 * (int)(oy + 1.f) + 9 * (int)(oz + 1.f);
}

// d = x_i - x_j - o.h for a stored image code (compute_dis, topology.py:9-10)
__device__ __forceinline__ void apply_shift(const MdgCell& c, int code, float& dx, float& dy,
                                            float& dz) {
    const float ox = (float)(code % 3 - 1), oy = (float)((code / 3) % 3 - 1),
                oz = (float)(code / 9 - 1);
    dx -= fmaf(oz, c.h[6], fmaf(oy, c.h[3], ox * c.h[0]));
    dy -= fmaf(oz, c.h[7], fmaf(oy, c.h[4], ox * c.h[1]));
    dz -= fmaf(oz, c.h[8], fmaf(oy, c.h[5], ox * c.h[2]));
}

// squared norm with the reference's association ((x^2 + y^2) + z^2), un-contracted, so the
// cutoff test selects the same pairs as torch's pow(2).sum(-1)
__device__ __forceinline__ float norm2_ref(float x, float y, float z) {
    return __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
}

// two-wide variants for the packed-math (v_pk_*_f32) pair loops
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 norm2_ref2(f32x2 x, f32x2 y, f32x2 z) {
#pragma clang fp contract(off)
    const f32x2 a = x * x;
    const f32x2 b = y * y;
    const f32x2 s = a + b;
    const f32x2 c = z * z;
    return s + c;
}
// orthorhombic minimum image of one component, two pairs at a time (see min_image)
__device__ __forceinline__ f32x2 min_image_diag2(f32x2 d, float inv, float h) {
    const f32x2 s = d * inv;
    const f32x2 o = {__builtin_amdgcn_fmed3f(rintf(s.x), -1.f, 1.f), __builtin_amdgcn_fmed3f(rintf(s.y), -1.f, 1.f)};
    return d - o * h;
}

// The same for |d * inv| < 1.5 (every atom inside a window 1.5 cells wide): the clamp is then the identity and
// rint runs as two packed adds around 1.5 * 2^23 (round-to-nearest-even of s to an integer, exactly rint(s)) --
// 4 packed instructions instead of 6 scalar + packed ones, bit-identical to min_image_diag2 there.
// (s = d * inv is rounded before the rint like the reference's matrix product: folding it into an fma would
//  pick the other image for a component within one ulp of half the cell -- measured: 3e-7 of the pairs.)
__device__ __forceinline__ f32x2 min_image_diag2_near(f32x2 d, float inv, float h) {
    const float M = 12582912.f;
    f32x2 o;
    {
#pragma clang fp contract(off)
        const f32x2 s = d * inv;                 // (rounded on its own: no fma with the add below)
        o = (s + M) - M;
    }
    return d - o * h;
}

// ----------------------------------------------------------------------------- pair forms
__device__ __forceinline__ float ipow(float x, int n) {
    float r = 1.f;
    while (n > 0) { if (n & 1) r *= x; x *= x; n >>= 1; }
    return r;
}

// LEVEL 0: u            1: + du, du_dtheta           2: + d2u, ddu_dtheta
struct PairOut {
    float u, du, d2u;
    float du_dth[MDG_MAX_THETA];
    float ddu_dth[MDG_MAX_THETA];
    // MDG_PAIR_TABLE only: first node of the bracketing cell and the four Hermite basis values there
    // (d phi'/d node = r * basis: the table gradient is a scatter, not a du_dth entry)
    int tg;
    float tb[4];
};

// Per-term constants hoisted out of the pair loop (parameters live in global memory).
struct TermConst {
    int kind, p, q;
    float c, rc2;
    float k0, k1, k2, k3, k4;   // LJ: sig, eps, 1/sig | Morse: a, phi, A, 1/(1+A) | Buck: A,B,C | Yukawa: eps,kappa
                                // Table: k0 = u0, k1 = 1/du, k2 = nodes - 1 - eps
    const float* tab;           // Table: [2 p] nodes (c1_g, du * dc1/du_g) in global memory (L1/L2 resident)
};

__device__ __forceinline__ TermConst term_prepare(const MdgPairTerm& t, const float* __restrict__ theta) {
    TermConst c;
    c.kind = t.kind; c.p = t.p; c.q = t.q; c.c = t.c; c.rc2 = t.cutoff * t.cutoff;
    c.k0 = c.k1 = c.k2 = c.k3 = c.k4 = 0.f;
    c.tab = nullptr;
    const float* th = theta + t.theta_off;
    switch (t.kind) {
    case MDG_PAIR_TABLE:
        c.k0 = t.a; c.k1 = 1.f / t.phi; c.k2 = (float)(t.p - 1) - 1e-3f; c.tab = th;
        break;
    case MDG_PAIR_LJ: c.k0 = th[0]; c.k1 = th[1]; c.k2 = 1.0f / th[0]; break;
    case MDG_PAIR_MORSE: {
        c.k0 = t.a; c.k1 = t.phi;
        c.k2 = t.phi >= 0.f ? 0.f : (expf(2.f * t.a / t.phi) - 2.f * expf(t.a / t.phi));
        c.k3 = 1.f / (1.f + c.k2);
    } break;
    case MDG_PAIR_BUCK: c.k0 = th[0]; c.k1 = th[1]; c.k2 = th[2]; break;
    default: c.k0 = th[0]; c.k1 = th[1]; break;
    }
    return c;
}

// compile-time kind of the single-term kernels: MDG_PAIR_LJ with the exponents fixed to 12-6
constexpr int KIND_LJ126 = 16;
constexpr int KIND_TABLE = 17;         // MDG_PAIR_TABLE in the fused small-system kernels
// trainable parameters per functional form (the theta slots pair_eval fills)
__host__ __device__ constexpr int kind_ntheta(int kind) {
    return kind == MDG_PAIR_MORSE ? 0 : (kind == MDG_PAIR_BUCK ? 3 : 2);
}

// phi and derivatives at squared distance d2.  r = d2 * rsq(d2), 1/r = rsq(d2) (v_rsq_f32, 1 ulp).
// KIND >= 0 fixes the functional form at compile time (single-term specialisations);
// KIND < 0 dispatches on tc.kind at run time.
template <int LEVEL, int KIND = -1>
__device__ __forceinline__ void pair_eval(const TermConst& tc, float d2, float& r, float& ir, PairOut& o) {
    ir = __builtin_amdgcn_rsqf(d2);
    r = d2 * ir;
    constexpr bool F126 = KIND == KIND_LJ126;
    switch (F126 ? MDG_PAIR_LJ : (KIND >= 0 ? KIND : tc.kind)) {
    case MDG_PAIR_LJ: {
        const float sig = tc.k0, eps = tc.k1, isig = tc.k2;
        const float s = sig * ir;
        float sp, sq;
        if (F126 || (tc.p == 12 && tc.q == 6)) { const float s2 = s * s; sq = s2 * s2 * s2; sp = sq * sq; }
        else { sp = ipow(s, tc.p); sq = ipow(s, tc.q); }
        sq *= tc.c;
        const float fp = F126 ? 12.f : (float)tc.p, fq = F126 ? 6.f : (float)tc.q;
        o.u = 4.f * eps * (sp - sq);
        if (LEVEL >= 1) {
            const float m1 = fq * sq - fp * sp;
            o.du = 4.f * eps * m1 * ir;
            o.du_dth[0] = -4.f * eps * m1 * isig;
            o.du_dth[1] = 4.f * (sp - sq);
            if (LEVEL >= 2) {
                o.d2u = 4.f * eps * (fp * (fp + 1.f) * sp - fq * (fq + 1.f) * sq) * ir * ir;
                o.ddu_dth[0] = 4.f * eps * (fq * fq * sq - fp * fp * sp) * ir * isig;
                o.ddu_dth[1] = 4.f * m1 * ir;
            }
        }
    } break;
    case MDG_PAIR_MORSE: {
        const float a = tc.k0, ph = tc.k1, A = tc.k2, inv = tc.k3;
        const float rp = powf(r, ph);
        const float x = a * (1.f - rp) / ph;
        const float e1 = expf(x), e2 = e1 * e1;
        o.u = (e2 - 2.f * e1 - A) * inv;
        if (LEVEL >= 1) {
            const float ux = (2.f * e2 - 2.f * e1) * inv;
            const float xr = -a * rp * ir;
            o.du = ux * xr;
            if (LEVEL >= 2) {
                const float uxx = (4.f * e2 - 2.f * e1) * inv;
                const float xrr = -a * (ph - 1.f) * rp * ir * ir;
                o.d2u = uxx * xr * xr + ux * xrr;
            }
        }
    } break;
    case MDG_PAIR_BUCK: {
        const float A = tc.k0, B = tc.k1, C = tc.k2;
        const float e = expf(-B * r);
        const float ir2 = ir * ir, ir6 = ir2 * ir2 * ir2;
        o.u = A * e - C * ir6;
        if (LEVEL >= 1) {
            o.du = -A * B * e + 6.f * C * ir6 * ir;
            o.du_dth[0] = e; o.du_dth[1] = -A * r * e; o.du_dth[2] = -ir6;
            if (LEVEL >= 2) {
                o.d2u = A * B * B * e - 42.f * C * ir6 * ir2;
                o.ddu_dth[0] = -B * e; o.ddu_dth[1] = A * e * (B * r - 1.f);
                o.ddu_dth[2] = 6.f * ir6 * ir;
            }
        }
    } break;
    case MDG_PAIR_TABLE: {
        // c1(u) = phi'(r)/r and its u-derivative from the cubic-Hermite table (see traj_small.hip,
        // force_table_packed): phi' = c1 r, phi'' = 2 u dc1/du + c1
        const float tt = fminf(fmaxf((d2 - tc.k0) * tc.k1, 0.f), tc.k2);
        const int g = (int)tt;
        const float fr = tt - (float)g;
        const float2 n0 = *reinterpret_cast<const float2*>(tc.tab + 2 * g);
        const float2 n1 = *reinterpret_cast<const float2*>(tc.tab + 2 * g + 2);
        const float om = 1.f - fr, fr2 = fr * fr, om2 = om * om;
        o.tg = g;
        o.tb[0] = (1.f + 2.f * fr) * om2; o.tb[1] = fr * om2; o.tb[2] = fr2 * (3.f - 2.f * fr); o.tb[3] = fr2 * (fr - 1.f);
        const float c1 = o.tb[0] * n0.x + o.tb[1] * n0.y + o.tb[2] * n1.x + o.tb[3] * n1.y;
        o.u = 0.f;                                   // (the energy is not tabulated: forces and HVP only)
        if (LEVEL >= 1) {
            o.du = c1 * r;
            if (LEVEL >= 2) {
                const float c1u = (6.f * fr * (fr - 1.f) * (n0.x - n1.x) + ((3.f * fr - 4.f) * fr + 1.f) * n0.y +
                                   (3.f * fr - 2.f) * fr * n1.y) * tc.k1;
                o.d2u = 2.f * c1u * d2 + c1;
            }
        }
#pragma unroll
        for (int k = 0; k < MDG_MAX_THETA; ++k) { o.du_dth[k] = 0.f; o.ddu_dth[k] = 0.f; }
    } break;
    default: {  // MDG_PAIR_YUKAWA
        const float eps = tc.k0, kap = tc.k1;
        const float e = expf(-kap * r);
        o.u = eps * e * ir;
        if (LEVEL >= 1) {
            const float kr1 = kap * r + 1.f;
            o.du = -eps * e * kr1 * ir * ir;
            o.du_dth[0] = e * ir; o.du_dth[1] = -eps * e;
            if (LEVEL >= 2) {
                o.d2u = eps * e * (kap * kap * r * r + 2.f * kap * r + 2.f) * ir * ir * ir;
                o.ddu_dth[0] = -e * kr1 * ir * ir; o.ddu_dth[1] = eps * e * kap;
            }
        }
    } break;
    }
}

// Workgroups are dispatched round-robin over the 8 XCDs, each with its own 4 MB L2: with the identity mapping
// neighbouring atoms -- which gather the same h rows -- land on different XCDs and every L2 sees the whole
// feature matrix.  xcd_chunk() gives XCD x the x-th contiguous eighth of the blocks instead.
__device__ __forceinline__ int xcd_chunk(int bid, int nblocks) {
    const int per = nblocks >> 3;
    if (per == 0 || bid >= (per << 3)) return bid;              // tail blocks keep their index
    return (bid & 7) * per + (bid >> 3);
}

// ----------------------------------------------------------------------------- reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sum across a contiguous power-of-two lane group of width W (W <= 64)
template <int W>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// run-time width variant (w is wave-uniform)
__device__ __forceinline__ float group_sum_rt(float v, int w) {
    for (int o = w >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- lane groups of 16 (one DPP row): sums without the LDS crossbar
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row, on every lane of the row (row_ror:8, row_ror:4, quad_perm [2,3,0,1], [1,0,3,2])
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_get<0x128>(v);
    v += dpp_get<0x124>(v);
    v += dpp_get<0x4E>(v);
    v += dpp_get<0xB1>(v);
    return v;
}
// ... then across the four rows: v_permlane16_swap / v_permlane32_swap hand each half its partner's sum
__device__ __forceinline__ float wave_sum_rows(float v) {
    v = row16_sum(v);
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]);
    const unsigned w = __builtin_bit_cast(unsigned, v);
    const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
}

// Block-wide sum, result broadcast to every thread.  `red` = LDS scratch of >= 17 floats.
// Deterministic (fixed tree).  Contains two barriers.
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();                       // protect `red` from the previous use
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float s = 0.f;
    for (int k = 0; k < nw; ++k) s += red[k];
    return s;
}

// Block-wide sum of NV values at once (one barrier pair).  `red` >= (#waves * NV) floats.
template <int NV>
__device__ __forceinline__ void block_sum_n(float (&v)[NV], float* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) red[wid * NV + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float s = 0.f;
        for (int w = 0; w < nw; ++w) s += red[w * NV + k];
        v[k] = s;
    }
}
