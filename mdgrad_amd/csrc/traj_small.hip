// Fused whole-trajectory kernels for small systems (K5-K7 of SURVEY 2.3).
//
// One workgroup integrates one replica for ALL steps with the state resident in LDS:
// the only HBM traffic is the initial state, the saved frames and (adjoint) the incoming
// frame gradients.  Forces are all-pairs minimum image over the LDS-resident positions,
// TPA lanes cooperating on one atom and combining with wave shuffles (no atomics, fixed
// summation order => bitwise reproducible).  Semantics follow the reference exactly:
//   forward  torchmd/sovlers.py:110-127 (NHverlet_update) / :25-40 (verlet_update),
//            RHS torchmd/md.py:210-240 (NoseHooverChain) / :133-150 (NVE)
//   adjoint  torchmd/sovlers.py:211-293 with backward branches :129-164 / :42-101
//            (vjp of the RHS written out analytically)
#include "common.hpp"

namespace {

struct TrajArgs {
    MdgTrajParams prm;
    MdgCell cell;
    MdgTerms terms;
    const float* theta;
    const float* mass;
    const float* t;
    const float* v0; const float* q0; const float* pv0;      // fwd inputs
    float* v_t; float* q_t; float* pv_t;                      // fwd outputs / adj inputs
    const float* g_v; const float* g_q; const float* g_pv;    // adj inputs
    float* adj_v0; float* adj_q0; float* adj_pv0; float* adj_theta;
    int32_t* nonfinite;
    int ld;                 // leading dimension of the SoA [3][ld] LDS arrays: 128 for N <= 128 (compile-time
                            // offsets, 8-byte pair loads in the packed loops), N rounded up to even otherwise
    // topology_update_freq > 1 (torchmd/md.py:200-204): the neighbour lists are rebuilt only at the calls whose running
    // count is a multiple of `freq` and are STALE in between -- pair set and image flags frozen at the rebuild positions,
    // no cutoff re-test (compute_dis over the stored list, topology.py:5-12).  code[rep][i][j] = 0 (no pair) or
    // (term bits << 5) | image code; persistent across launches, like the reference's nbr_list / offsets attributes.
    uint16_t* code;
    int freq;
    long long count0;       // value of the integrator's update_count at the first force call of this launch
};

constexpr int KMAX_ALL = MDG_MAX_TERMS * MDG_MAX_THETA;
constexpr int RED_FLOATS = 16 * (KMAX_ALL + 2);
// Kernels are specialised on <DIAG, NT, KIND>: NT = compile-time bound on the number of pair
// terms (1 or MDG_MAX_TERMS), KIND = the functional form when NT == 1 (-1 = run-time switch).
// The lane-group width TPA is a run-time power of two.
#define KMAX (NT * MDG_MAX_THETA)

// LJ 12-6 (times the q-term coefficient c), orthorhombic cell, no mask: two pairs per lane and iteration
// in packed fp32 (v_pk_fma/mul/add_f32), written in even powers of 1/r only -- one v_rcp_f32 per pair, no
// square root:   s2 = sig^2/d2, s6 = s2^3, s12 = s6^2
//   phi'/r        = 4 eps (6 c s6 - 12 s12) / d2                                  =: c1
//   phi''         = 4 eps (156 s12 - 42 c s6) / d2
//   -H_ij w_ij    = -[(phi'' - phi'/r) (D.w_ij) D / d2 + (phi'/r) w_ij]           (D = x_j - x_i)
//   d(w.F)/dsig  += 1/2 4 eps (36 c s6 - 144 s12) (D.w_ij) / (sig d2)
//   d(w.F)/deps  += 2 (6 c s6 - 12 s12) (D.w_ij) / d2
// (the same quantities force_all_pairs gets from pair_eval<LEVEL, MDG_PAIR_LJ> through r and 1/r).
// A lane takes the CONSECUTIVE neighbours (j, j+1): with the SoA rows 8-byte aligned and of even length one
// ds_read_b64 per component delivers both operands of the packed arithmetic; LDC = 128 (every N <= 128)
// turns the row offsets into instruction immediates -- 6 LDS instructions and one address per iteration
// instead of 12 + 12.  (Row padding is zero-filled, so the odd-N tail reads finite values.)
template <int LEVEL, int LDC, bool EVEN, bool NEAR>
__device__ __forceinline__ void force_lj126_packed(const TrajArgs& A, int tpa_log2, const float* __restrict__ q,
                                                   const float* __restrict__ w, float* __restrict__ f,
                                                   float* __restrict__ dq, float& th_sig, float& th_eps, bool th_on) {
    const int N = A.prm.n_atoms, LD = LDC ? LDC : A.ld;
    const int TPA = 1 << tpa_log2;
    const int slots = blockDim.x >> tpa_log2;
    const int slot = threadIdx.x >> tpa_log2, sub = threadIdx.x & (TPA - 1);
    const TermConst t0 = term_prepare(A.terms.t[0], A.theta);
    const float sig2 = t0.k0 * t0.k0, e4 = 4.f * t0.k1, cq = t0.c, rc2 = t0.rc2;
    // 4 eps and the q-term coefficient folded into the polynomial coefficients (A = s6, B = s12)
    const float m1a = 6.f * e4 * cq, m1b = 12.f * e4;                 // phi'/r  = (m1a A - m1b B) / d2
    const float ka = (42.f + 6.f) * e4 * cq, kb = (156.f + 12.f) * e4;   // phi'' - phi'/r = (kb B - ka A) / d2
    const float tsa = 18.f * e4 * t0.k2 * cq, tsb = 72.f * e4 * t0.k2;   // 1/2 d(phi'/r)/dsig r-part
    const float tea = 12.f * cq, teb = 24.f;                          // 2 m1 / (4 eps)
    const float ivx = A.cell.inv[0], ivy = A.cell.inv[4], ivz = A.cell.inv[8];
    const float hx = A.cell.h[0], hy = A.cell.h[4], hz = A.cell.h[8];
    // theta gradients: both are linear in  S6 = sum s6 (w.D)/d2  and  S12 = sum s12 (w.D)/d2, so the loop
    // carries those two sums only (2 packed fma per pair-of-pairs, no branch on th_on)
    f32x2 S6 = {0.f, 0.f}, S12 = {0.f, 0.f};
    for (int i = slot; i < N; i += slots) {
        const float xi = q[i], yi = q[LD + i], zi = q[2 * LD + i];
        float wxi = 0.f, wyi = 0.f, wzi = 0.f;
        if (LEVEL >= 2) { wxi = w[i]; wyi = w[LD + i]; wzi = w[2 * LD + i]; }
        f32x2 fx = {0.f, 0.f}, fy = fx, fz = fx, gx = fx, gy = fx, gz = fx;
#pragma unroll 2
        for (int j = 2 * sub; j < N; j += 2 * TPA) {
            const bool live2 = EVEN || j + 1 < N;           // (EVEN: N is even, every lane pair is live)
            const float* qj = q + j;
            const f32x2 qx = *reinterpret_cast<const f32x2*>(qj), qy = *reinterpret_cast<const f32x2*>(qj + LD),
                        qz = *reinterpret_cast<const f32x2*>(qj + 2 * LD);
            f32x2 dx = qx - xi, dy = qy - yi, dz = qz - zi;                        // D = x_j - x_i
            f32x2 ax = {0.f, 0.f}, ay = ax, az = ax;
            if (LEVEL >= 2) {
                const float* wj = w + j;
                ax = wxi - *reinterpret_cast<const f32x2*>(wj); ay = wyi - *reinterpret_cast<const f32x2*>(wj + LD);
                az = wzi - *reinterpret_cast<const f32x2*>(wj + 2 * LD);
            }
            if constexpr (NEAR) {
                dx = min_image_diag2_near(dx, ivx, hx); dy = min_image_diag2_near(dy, ivy, hy);
                dz = min_image_diag2_near(dz, ivz, hz);
            } else {
                dx = min_image_diag2(dx, ivx, hx); dy = min_image_diag2(dy, ivy, hy); dz = min_image_diag2(dz, ivz, hz);
            }
            const f32x2 d2 = norm2_ref2(dx, dy, dz);
            const bool ok0 = (d2.x != 0.f) && (d2.x < rc2);                         // topology.py:67
            const bool ok1 = live2 && (d2.y != 0.f) && (d2.y < rc2);
            // 1/d2 selected to 0 for a rejected pair (the rcp of a self pair's 0 is never used): s6, s12 and
            // everything below are then exactly zero
            const f32x2 i2 = {ok0 ? __builtin_amdgcn_rcpf(d2.x) : 0.f, ok1 ? __builtin_amdgcn_rcpf(d2.y) : 0.f};
            const f32x2 s2 = sig2 * i2;
            const f32x2 s6 = s2 * s2 * s2;
            const f32x2 s12 = s6 * s6;
            const f32x2 c1 = (m1a * s6 - m1b * s12) * i2;
            fx += c1 * dx; fy += c1 * dy; fz += c1 * dz;      // F_i += (phi'/r) D
            if (LEVEL >= 2) {
                const f32x2 b = dx * ax + dy * ay + dz * az;
                const f32x2 bi = b * i2;                      // (w.D) / d2
                const f32x2 k2 = (kb * s12 - ka * s6) * (bi * i2);   // (phi'' - phi'/r) (w.D) / d2
                gx += k2 * dx; gx += c1 * ax;                 // (accumulated with the opposite sign: two fma per
                gy += k2 * dy; gy += c1 * ay;                 //  component; negated once after the reduction)
                gz += k2 * dz; gz += c1 * az;
                S6 += s6 * bi; S12 += s12 * bi;
            }
        }
        float sx = group_sum_rt(fx.x + fx.y, TPA), sy = group_sum_rt(fy.x + fy.y, TPA), sz = group_sum_rt(fz.x + fz.y, TPA);
        float ux = 0.f, uy = 0.f, uz = 0.f;
        if (LEVEL >= 2) {
            ux = -group_sum_rt(gx.x + gx.y, TPA); uy = -group_sum_rt(gy.x + gy.y, TPA);
            uz = -group_sum_rt(gz.x + gz.y, TPA);
        }
        if (sub == 0) {
            f[i] = sx; f[LD + i] = sy; f[2 * LD + i] = sz;
            if (LEVEL >= 2) { dq[i] = ux; dq[LD + i] = uy; dq[2 * LD + i] = uz; }
        }
    }
    if (LEVEL >= 2 && th_on) {                                // (the first NHC evaluation of an interval skips it)
        const float a6 = S6.x + S6.y, a12 = S12.x + S12.y;
        th_sig += tsa * a6 - tsb * a12;
        th_eps += tea * a6 - teb * a12;
    }
}

// Tabulated pair model (MDG_PAIR_TABLE): c1(u) = phi'(r)/r on a uniform grid in u = r^2, cubic-Hermite
// nodes (value, du * slope) resident in LDS.  Force = c1 D; Hessian term (phi'' - phi'/r)/r^2 = 2 dc1/du is
// the derivative of the SAME interpolant, so force and Hessian-vector product stay consistent.
// The parameter vjp is the gradient w.r.t. the table nodes: d(w.F)/dnode += 1/2 (D.w_ij) basis_node(u) per
// directed pair, scattered into LDS.  Float LDS atomics are ~13x slower than integer ones on gfx950
// (tools/micro/lds_atomics.hip), so the scatter is fixed point in one int64 word per entry (fx64, common.hpp; round 6 --
// two int32 planes before, whose sums could wrap unnoticed): order-independent => bitwise reproducible, like every
// other reduction here.
struct TableRef {
    const float2* tab;      // [M] (c1_g, du * dc1/du_g)
    unsigned long long* g64;   // [2M] or nullptr (no accumulation in this evaluation)
    float gw;               // weight of this evaluation's contributions: 1/2 * h * 2^S
};

__device__ __forceinline__ void table_scatter(const TableRef& T, int idx, float val, float& vmax) {
    vmax = val == val ? fmaxf(vmax, fabsf(val)) : __builtin_inff();       // (a NaN contribution counts as out of range)
    atomicAdd(T.g64 + idx, fx64(val));
}

template <int LEVEL, int LDC, bool NEAR>
__device__ __forceinline__ void force_table_packed(const TrajArgs& A, int tpa_log2, const float* __restrict__ q,
                                                   const float* __restrict__ w, float* __restrict__ f,
                                                   float* __restrict__ dq, const TableRef& T, float& vmax) {
    const int N = A.prm.n_atoms, LD = LDC ? LDC : A.ld;
    const int TPA = 1 << tpa_log2;
    const int slots = blockDim.x >> tpa_log2;
    const int slot = threadIdx.x >> tpa_log2, sub = threadIdx.x & (TPA - 1);
    const MdgPairTerm& t0 = A.terms.t[0];
    const float rc2 = t0.cutoff * t0.cutoff, u0 = t0.a, inv_du = 1.f / t0.phi;
    const float tmax = (float)(t0.p - 1) - 1e-3f;
    const float ivx = A.cell.inv[0], ivy = A.cell.inv[4], ivz = A.cell.inv[8];
    const float hx = A.cell.h[0], hy = A.cell.h[4], hz = A.cell.h[8];
    const bool acc = LEVEL >= 2 && T.g64 != nullptr;
    for (int i = slot; i < N; i += slots) {
        const float xi = q[i], yi = q[LD + i], zi = q[2 * LD + i];
        float wxi = 0.f, wyi = 0.f, wzi = 0.f;
        if (LEVEL >= 2) { wxi = w[i]; wyi = w[LD + i]; wzi = w[2 * LD + i]; }
        f32x2 fx = {0.f, 0.f}, fy = fx, fz = fx, gx = fx, gy = fx, gz = fx;
        for (int j = 2 * sub; j < N; j += 2 * TPA) {                             // consecutive pair (j, j+1): see force_lj126_packed
            const bool live2 = NEAR || j + 1 < N;                                 // (NEAR implies an even N)
            const float* qj = q + j;
            f32x2 dx = *reinterpret_cast<const f32x2*>(qj) - xi, dy = *reinterpret_cast<const f32x2*>(qj + LD) - yi,
                  dz = *reinterpret_cast<const f32x2*>(qj + 2 * LD) - zi;          // D = x_j - x_i
            f32x2 ax = {0.f, 0.f}, ay = ax, az = ax;
            if (LEVEL >= 2) {
                const float* wj = w + j;
                ax = wxi - *reinterpret_cast<const f32x2*>(wj); ay = wyi - *reinterpret_cast<const f32x2*>(wj + LD);
                az = wzi - *reinterpret_cast<const f32x2*>(wj + 2 * LD);
            }
            if constexpr (NEAR) {
                dx = min_image_diag2_near(dx, ivx, hx); dy = min_image_diag2_near(dy, ivy, hy);
                dz = min_image_diag2_near(dz, ivz, hz);
            } else {
                dx = min_image_diag2(dx, ivx, hx); dy = min_image_diag2(dy, ivy, hy); dz = min_image_diag2(dz, ivz, hz);
            }
            const f32x2 d2 = norm2_ref2(dx, dy, dz);
            const bool ok0 = (d2.x != 0.f) && (d2.x < rc2);                         // topology.py:67
            const bool ok1 = live2 && (d2.y != 0.f) && (d2.y < rc2);
            const f32x2 sel = {ok0 ? 1.f : 0.f, ok1 ? 1.f : 0.f};
            // a live pair below the first node: the table does not cover it (the forward kernel reports it through
            // the per-replica flag, bit 1, and the host raises instead of returning clamped forces)
            if (LEVEL == 1 && ((ok0 && d2.x < u0) || (ok1 && d2.y < u0))) vmax = 1.f;
            // grid coordinate (clamped: below the first node the first cell is extrapolated with fr = 0)
            f32x2 tt = (d2 - u0) * inv_du;
            tt.x = fminf(fmaxf(ok0 ? tt.x : 0.f, 0.f), tmax); tt.y = fminf(fmaxf(ok1 ? tt.y : 0.f, 0.f), tmax);
            const int g0 = (int)tt.x, g1 = (int)tt.y;
            const f32x2 fr = {tt.x - (float)g0, tt.y - (float)g1};
            const float2 a0 = T.tab[g0], b0 = T.tab[g0 + 1], a1 = T.tab[g1], b1 = T.tab[g1 + 1];
            const f32x2 v0 = {a0.x, a1.x}, s0 = {a0.y, a1.y}, v1 = {b0.x, b1.x}, s1 = {b0.y, b1.y};
            const f32x2 om = 1.f - fr, fr2 = fr * fr, om2 = om * om;
            const f32x2 h00 = (1.f + 2.f * fr) * om2, h10 = fr * om2, h01 = fr2 * (3.f - 2.f * fr), h11 = fr2 * (fr - 1.f);
            const f32x2 c1 = (h00 * v0 + h10 * s0 + h01 * v1 + h11 * s1) * sel;
            fx += c1 * dx; fy += c1 * dy; fz += c1 * dz;      // F_i += (phi'/r) D
            if (LEVEL >= 2) {
                const f32x2 b = dx * ax + dy * ay + dz * az;
                const f32x2 e00 = 6.f * fr * (fr - 1.f), e10 = (3.f * fr - 4.f) * fr + 1.f, e11 = (3.f * fr - 2.f) * fr;
                const f32x2 c1u = (e00 * (v0 - v1) + e10 * s0 + e11 * s1) * (inv_du * sel);
                const f32x2 k2 = (2.f * c1u) * b;
                gx -= k2 * dx + c1 * ax;
                gy -= k2 * dy + c1 * ay;
                gz -= k2 * dz + c1 * az;
                if (acc) {
                    const f32x2 x = (T.gw * b) * sel;
                    if (ok0) {
                        table_scatter(T, 2 * g0, x.x * h00.x, vmax); table_scatter(T, 2 * g0 + 1, x.x * h10.x, vmax);
                        table_scatter(T, 2 * g0 + 2, x.x * h01.x, vmax); table_scatter(T, 2 * g0 + 3, x.x * h11.x, vmax);
                    }
                    if (ok1) {
                        table_scatter(T, 2 * g1, x.y * h00.y, vmax); table_scatter(T, 2 * g1 + 1, x.y * h10.y, vmax);
                        table_scatter(T, 2 * g1 + 2, x.y * h01.y, vmax); table_scatter(T, 2 * g1 + 3, x.y * h11.y, vmax);
                    }
                }
            }
        }
        float sx = group_sum_rt(fx.x + fx.y, TPA), sy = group_sum_rt(fy.x + fy.y, TPA), sz = group_sum_rt(fz.x + fz.y, TPA);
        float ux = 0.f, uy = 0.f, uz = 0.f;
        if (LEVEL >= 2) {
            ux = group_sum_rt(gx.x + gx.y, TPA); uy = group_sum_rt(gy.x + gy.y, TPA); uz = group_sum_rt(gz.x + gz.y, TPA);
        }
        if (sub == 0) {
            f[i] = sx; f[LD + i] = sy; f[2 * LD + i] = sz;
            if (LEVEL >= 2) { dq[i] = ux; dq[LD + i] = uy; dq[2 * LD + i] = uz; }
        }
    }
}

// All-pairs force (LEVEL 1) or force + Hessian-vector product + parameter vjp (LEVEL 2).
//   f   [3][N] <-  F = -dU/dq
//   dq  [3][N] <-  d(w.F)/dq = -H w                      (LEVEL 2)
//   dth [K]    +=  per-thread partial of d(w.F)/dtheta   (LEVEL 2; caller block-reduces)
template <bool DIAG, int NT, int KIND, int LEVEL>
__device__ __forceinline__ void force_all_pairs(const TrajArgs& A, int tpa_log2, const float* __restrict__ q,
                                                const float* __restrict__ w, float* __restrict__ f,
                                                float* __restrict__ dq, float (&dth)[KMAX], const TableRef& TB,
                                                float& vmax, bool th_on = true) {
    const int N = A.prm.n_atoms, LD = A.ld;
    // fast variant of the packed loops: even N and every atom within [-0.24, 1.24] cell lengths (one block-wide
    // vote per evaluation; positions are wrapped at every epoch, md.py:66, so a trajectory leaves the window
    // only after drifting a quarter cell), else the general one
    bool near = false;
    if constexpr (KIND == KIND_TABLE || KIND == KIND_LJ126) {
        bool out = false;
        for (int c = 0; c < 3; ++c) {
            const float iv = A.cell.inv[4 * c];
            for (int i = threadIdx.x; i < N; i += blockDim.x) {
                const float sc = q[c * LD + i] * iv;
                out |= !(sc > -0.24f && sc < 1.24f);
            }
        }
        near = !(N & 1) && !__syncthreads_or(out);
    }
    if constexpr (KIND == KIND_TABLE) {
        if (A.ld == 128) {
            if (near) force_table_packed<LEVEL, 128, true>(A, tpa_log2, q, w, f, dq, TB, vmax);
            else force_table_packed<LEVEL, 128, false>(A, tpa_log2, q, w, f, dq, TB, vmax);
        } else {
            if (near) force_table_packed<LEVEL, 0, true>(A, tpa_log2, q, w, f, dq, TB, vmax);
            else force_table_packed<LEVEL, 0, false>(A, tpa_log2, q, w, f, dq, TB, vmax);
        }
        return;
    }
    const int TPA = 1 << tpa_log2;
    const int slots = blockDim.x >> tpa_log2;
    const int slot = threadIdx.x >> tpa_log2, sub = threadIdx.x & (TPA - 1);
    const int nt = NT == 1 ? 1 : A.terms.n_terms;
    TermConst tc[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m)
        if (m < nt) tc[m] = term_prepare(A.terms.t[m], A.theta);
    if constexpr (KIND == KIND_LJ126) {
        if (A.ld == 128) {
            if (near) force_lj126_packed<LEVEL, 128, true, true>(A, tpa_log2, q, w, f, dq, dth[0], dth[1], th_on);
            else force_lj126_packed<LEVEL, 128, false, false>(A, tpa_log2, q, w, f, dq, dth[0], dth[1], th_on);
        } else {
            if (near) force_lj126_packed<LEVEL, 0, true, true>(A, tpa_log2, q, w, f, dq, dth[0], dth[1], th_on);
            else force_lj126_packed<LEVEL, 0, false, false>(A, tpa_log2, q, w, f, dq, dth[0], dth[1], th_on);
        }
        return;
    }
    if constexpr (KIND >= 0) {
        // single unmasked term of a fixed form: branch-free body (a rejected pair is evaluated at the
        // cutoff and multiplied by zero), so the unrolled iterations interleave and every LDS operand of a
        // pair is fetched in one round trip
        constexpr int NTH = kind_ntheta(KIND == KIND_LJ126 ? MDG_PAIR_LJ : KIND);
        const TermConst t0 = tc[0];
        for (int i = slot; i < N; i += slots) {
            const float xi = q[i], yi = q[LD + i], zi = q[2 * LD + i];
            float wxi = 0.f, wyi = 0.f, wzi = 0.f;
            if (LEVEL >= 2) { wxi = w[i]; wyi = w[LD + i]; wzi = w[2 * LD + i]; }
            float fx = 0.f, fy = 0.f, fz = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll 2
            for (int j = sub; j < N; j += TPA) {
                float dx = q[j] - xi, dy = q[LD + j] - yi, dz = q[2 * LD + j] - zi;   // D = x_j - x_i
                float ax = 0.f, ay = 0.f, az = 0.f;
                if (LEVEL >= 2) { ax = wxi - w[j]; ay = wyi - w[LD + j]; az = wzi - w[2 * LD + j]; }
                min_image<DIAG>(A.cell, dx, dy, dz);
                const float d2 = norm2_ref(dx, dy, dz);
                const bool ok = (d2 != 0.f) && (d2 < t0.rc2);                       // topology.py:67
                PairOut o;
                float r, ir;
                pair_eval<LEVEL, KIND>(t0, ok ? d2 : t0.rc2, r, ir, o);
                const float c1 = ok ? o.du * ir : 0.f;       // F_i += phi' * D / r   (rhat = -D/r)
                fx = fmaf(c1, dx, fx); fy = fmaf(c1, dy, fy); fz = fmaf(c1, dz, fz);
                if (LEVEL >= 2) {
                    const float rx = -dx * ir, ry = -dy * ir, rz = -dz * ir;
                    const float a = rx * ax + ry * ay + rz * az;
                    const float c2 = (ok ? o.d2u : 0.f) * a - c1 * a, c3 = c1;
                    gx -= c2 * rx + c3 * ax;
                    gy -= c2 * ry + c3 * ay;
                    gz -= c2 * rz + c3 * az;
#pragma unroll
                    for (int k = 0; k < NTH; ++k) dth[k] -= 0.5f * (ok ? o.ddu_dth[k] : 0.f) * a;
                }
            }
            fx = group_sum_rt(fx, TPA); fy = group_sum_rt(fy, TPA); fz = group_sum_rt(fz, TPA);
            if (LEVEL >= 2) { gx = group_sum_rt(gx, TPA); gy = group_sum_rt(gy, TPA); gz = group_sum_rt(gz, TPA); }
            if (sub == 0) {
                f[i] = fx; f[LD + i] = fy; f[2 * LD + i] = fz;
                if (LEVEL >= 2) { dq[i] = gx; dq[LD + i] = gy; dq[2 * LD + i] = gz; }
            }
        }
        return;
    }
    for (int i = slot; i < N; i += slots) {
        const float xi = q[i], yi = q[LD + i], zi = q[2 * LD + i];
        float wxi = 0.f, wyi = 0.f, wzi = 0.f;
        if (LEVEL >= 2) { wxi = w[i]; wyi = w[LD + i]; wzi = w[2 * LD + i]; }
        float fx = 0.f, fy = 0.f, fz = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
        for (int j = sub; j < N; j += TPA) {
            float dx = q[j] - xi, dy = q[LD + j] - yi, dz = q[2 * LD + j] - zi;   // D = x_j - x_i
            min_image<DIAG>(A.cell, dx, dy, dz);
            const float d2 = norm2_ref(dx, dy, dz);
            if (d2 == 0.f) continue;                                           // topology.py:67
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                if (m >= nt) break;
                if (!(d2 < tc[m].rc2)) continue;
                const uint8_t* mk = A.terms.t[m].mask;
                if (mk && !mk[(size_t)i * N + j]) continue;
                PairOut o;
                float r, ir;
                pair_eval<LEVEL, KIND>(tc[m], d2, r, ir, o);
                const float c1 = o.du * ir;              // F_i += phi' * D / r   (rhat = -D/r)
                fx = fmaf(c1, dx, fx); fy = fmaf(c1, dy, fy); fz = fmaf(c1, dz, fz);
                if (LEVEL >= 2) {
                    const float rx = -dx * ir, ry = -dy * ir, rz = -dz * ir;
                    const float ax = wxi - w[j], ay = wyi - w[LD + j], az = wzi - w[2 * LD + j];
                    const float a = rx * ax + ry * ay + rz * az;
                    const float c2 = o.d2u * a - c1 * a, c3 = c1;
                    // hv = phi'' a rhat + (phi'/r)(wij - a rhat) = (phi'' - phi'/r) a rhat + (phi'/r) wij
                    gx -= c2 * rx + c3 * ax;
                    gy -= c2 * ry + c3 * ay;
                    gz -= c2 * rz + c3 * az;
#pragma unroll
                    for (int k = 0; k < MDG_MAX_THETA; ++k)
                        if (k < A.terms.t[m].n_theta) dth[m * MDG_MAX_THETA + k] -= 0.5f * o.ddu_dth[k] * a;
                }
            }
        }
        fx = group_sum_rt(fx, TPA); fy = group_sum_rt(fy, TPA); fz = group_sum_rt(fz, TPA);
        if (LEVEL >= 2) { gx = group_sum_rt(gx, TPA); gy = group_sum_rt(gy, TPA); gz = group_sum_rt(gz, TPA); }
        if (sub == 0) {
            f[i] = fx; f[LD + i] = fy; f[2 * LD + i] = fz;
            if (LEVEL >= 2) { dq[i] = gx; dq[LD + i] = gy; dq[2 * LD + i] = gz; }
        }
    }
}

// force_all_pairs for stale neighbour lists (topology_update_freq > 1; generic multi-term kernels only).  `rebuild`
// (workgroup-uniform): this call's count is a multiple of the frequency -- the lists of every term are searched at the
// current positions exactly as generate_nbr_list does (minimum image, un-contracted d^2 < rc^2, != 0, selection mask) and
// written to the replica's code matrix; otherwise the stored pairs are evaluated with their frozen image flags, whatever
// their distance is now (the reference's PairPotentials.forward over self.nbr_list / self.offsets, interface.py:298-300).
template <bool DIAG, int NT, int LEVEL>
__device__ __forceinline__ void force_pairs_stale(const TrajArgs& A, int tpa_log2, const float* __restrict__ q,
                                                  const float* __restrict__ w, float* __restrict__ f,
                                                  float* __restrict__ dq, float (&dth)[KMAX], bool rebuild) {
    const int N = A.prm.n_atoms, LD = A.ld;
    const int TPA = 1 << tpa_log2;
    const int slots = blockDim.x >> tpa_log2;
    const int slot = threadIdx.x >> tpa_log2, sub = threadIdx.x & (TPA - 1);
    const int nt = NT == 1 ? 1 : A.terms.n_terms;
    uint16_t* code = A.code + (size_t)blockIdx.x * N * N;
    TermConst tc[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m)
        if (m < nt) tc[m] = term_prepare(A.terms.t[m], A.theta);
    for (int i = slot; i < N; i += slots) {
        const float xi = q[i], yi = q[LD + i], zi = q[2 * LD + i];
        float wxi = 0.f, wyi = 0.f, wzi = 0.f;
        if (LEVEL >= 2) { wxi = w[i]; wyi = w[LD + i]; wzi = w[2 * LD + i]; }
        float fx = 0.f, fy = 0.f, fz = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
        for (int j = sub; j < N; j += TPA) {
            float dx = q[j] - xi, dy = q[LD + j] - yi, dz = q[2 * LD + j] - zi;   // D = x_j - x_i
            unsigned c;
            if (rebuild) {
                const int img = min_image<DIAG>(A.cell, dx, dy, dz);
                const float d2n = norm2_ref(dx, dy, dz);
                unsigned bits = 0;
                if (d2n != 0.f) {                                               // topology.py:67
#pragma unroll
                    for (int m = 0; m < NT; ++m) {
                        if (m >= nt) break;
                        const uint8_t* mk = A.terms.t[m].mask;
                        if (d2n < tc[m].rc2 && (!mk || mk[(size_t)i * N + j])) bits |= 1u << m;
                    }
                }
                c = bits ? ((bits << 5) | (unsigned)img) : 0u;
                code[(size_t)i * N + j] = (uint16_t)c;
            } else {
                c = code[(size_t)i * N + j];
                if (c) {                                                        // D += o . h with the stored flags
                    const int img = (int)(c & 31u);
                    const float ox = (float)(img % 3 - 1), oy = (float)((img / 3) % 3 - 1), oz = (float)(img / 9 - 1);
                    dx += fmaf(oz, A.cell.h[6], fmaf(oy, A.cell.h[3], ox * A.cell.h[0]));
                    dy += fmaf(oz, A.cell.h[7], fmaf(oy, A.cell.h[4], ox * A.cell.h[1]));
                    dz += fmaf(oz, A.cell.h[8], fmaf(oy, A.cell.h[5], ox * A.cell.h[2]));
                }
            }
            if (!c) continue;
            const float d2 = norm2_ref(dx, dy, dz);
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                if (m >= nt) break;
                if (!((c >> (5 + m)) & 1u)) continue;
                PairOut o;
                float r, ir;
                pair_eval<LEVEL, -1>(tc[m], d2, r, ir, o);
                const float c1 = o.du * ir;
                fx = fmaf(c1, dx, fx); fy = fmaf(c1, dy, fy); fz = fmaf(c1, dz, fz);
                if (LEVEL >= 2) {
                    const float rx = -dx * ir, ry = -dy * ir, rz = -dz * ir;
                    const float ax = wxi - w[j], ay = wyi - w[LD + j], az = wzi - w[2 * LD + j];
                    const float a = rx * ax + ry * ay + rz * az;
                    const float c2 = o.d2u * a - c1 * a, c3 = c1;
                    gx -= c2 * rx + c3 * ax;
                    gy -= c2 * ry + c3 * ay;
                    gz -= c2 * rz + c3 * az;
#pragma unroll
                    for (int k = 0; k < MDG_MAX_THETA; ++k)
                        if (k < A.terms.t[m].n_theta) dth[m * MDG_MAX_THETA + k] -= 0.5f * o.ddu_dth[k] * a;
                }
            }
        }
        fx = group_sum_rt(fx, TPA); fy = group_sum_rt(fy, TPA); fz = group_sum_rt(fz, TPA);
        if (LEVEL >= 2) { gx = group_sum_rt(gx, TPA); gy = group_sum_rt(gy, TPA); gz = group_sum_rt(gz, TPA); }
        if (sub == 0) {
            f[i] = fx; f[LD + i] = fy; f[2 * LD + i] = fz;
            if (LEVEL >= 2) { dq[i] = gx; dq[LD + i] = gy; dq[2 * LD + i] = gz; }
        }
    }
}

// the call with running index `e` of this launch rebuilds the lists (md.py:200-204: update_count % freq == 0)
__device__ __forceinline__ bool stale_due(const TrajArgs& A, long long e) { return (A.count0 + e) % (long long)A.freq == 0; }

// Nose-Hoover chain bath right-hand side, entry k (md.py:234-236)
// (Q is an LDS copy of prm.Q: indexing the by-value kernel argument with a run-time index
//  would force the whole argument struct into scratch)
__device__ __forceinline__ float bath_rhs(const TrajArgs& A, const float* Q, const float* pv, float ke, int k) {
    const int C = A.prm.n_chains;
    const float T = A.prm.T;
    if (k == 0) return 2.f * (ke - T * A.prm.n_dof * 0.5f) - pv[0] * pv[1] / Q[1];
    if (k == C - 1) return pv[C - 2] * pv[C - 2] / Q[C - 2] - T;
    return (pv[k - 1] * pv[k - 1] / Q[k - 1] - T) - pv[k + 1] * pv[k] / Q[k + 1];
}

// the 3N degrees of freedom of an SoA [3][LD] array without div/mod: e = c * LD + i
#define MDG_FOR_DOF(e, i, c)    \
    for (int c = 0; c < 3; ++c) \
        for (int i = threadIdx.x, e = c * LD + threadIdx.x; i < N; i += blockDim.x, e += blockDim.x)

// AoS [N,3] global  <->  SoA [3][N] LDS
__device__ __forceinline__ void load_soa(float* dst, const float* __restrict__ src, int N, int LD) {
    for (int e = threadIdx.x; e < 3 * N; e += blockDim.x) dst[(e % 3) * LD + e / 3] = src[e];
}
__device__ __forceinline__ void store_aos(float* __restrict__ dst, const float* src, int N, int LD) {
    for (int e = threadIdx.x; e < 3 * N; e += blockDim.x) dst[e] = src[(e % 3) * LD + e / 3];
}

// ------------------------------------------------------------------------------------ forward
template <bool DIAG, int NT, int KIND>
__global__ __launch_bounds__(1024) void traj_fwd_kernel(const TrajArgs A, const int tpa_log2) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N = A.prm.n_atoms, T = A.prm.n_frames, C = A.prm.n_chains, LD = A.ld;
    const bool nhc = A.prm.ensemble == 0;
    const int rep = blockIdx.x;
    float* q = smem;            // [3][LD]
    float* v = q + 3 * LD;      // [3][LD]
    float* vh = v + 3 * LD;     // [3][LD] half-step velocity increment
    float* f = vh + 3 * LD;     // [3][LD]
    float* ms = f + 3 * LD;     // [LD]
    float* pv = ms + LD;        // [C]
    for (int e = threadIdx.x; e < 13 * LD; e += blockDim.x) smem[e] = 0.f;      // padding columns stay finite
    __syncthreads();
    float* ph = pv + MDG_MAX_CHAINS;
    float* pb = ph + MDG_MAX_CHAINS;
    float* pvh = pb + MDG_MAX_CHAINS;   // [C] pv + ph
    float* Qs = pvh + MDG_MAX_CHAINS;   // [C] thermostat masses
    float* red = Qs + MDG_MAX_CHAINS;   // [RED_FLOATS]
    float dth_unused[KMAX];
    float vmax_unused = 0.f;
    TableRef TB{nullptr, nullptr, 0.f};
    if constexpr (KIND == KIND_TABLE) {                    // table nodes resident in LDS for the whole trajectory
        float2* tab = reinterpret_cast<float2*>(red + RED_FLOATS);       // (13 LD + chains + RED_FLOATS is even)
        const float* th = A.theta + A.terms.t[0].theta_off;
        for (int g = threadIdx.x; g < A.terms.t[0].p; g += blockDim.x) tab[g] = make_float2(th[2 * g], th[2 * g + 1]);
        TB.tab = tab;
    }
#pragma unroll
    for (int c = 0; c < MDG_MAX_CHAINS; ++c)
        if (threadIdx.x == c) Qs[c] = A.prm.Q[c];
    const float Q0 = A.prm.Q[0], iQ0 = 1.f / Q0;
    (void)Q0;   // (per-element divisions by the atom's mass and by Q0 run as v_rcp_f32 / a multiply: 1 ulp)

    load_soa(q, A.q0 + (size_t)rep * N * 3, N, LD);
    load_soa(v, A.v0 + (size_t)rep * N * 3, N, LD);
    for (int i = threadIdx.x; i < N; i += blockDim.x) ms[i] = A.mass[i];
    if (nhc && threadIdx.x < C) pv[threadIdx.x] = A.pv0[(size_t)rep * C + threadIdx.x];
    __syncthreads();
    // frame 0 = inputs (tinydiffeq.py:63)
    store_aos(A.q_t + ((size_t)rep * T) * N * 3, q, N, LD);
    store_aos(A.v_t + ((size_t)rep * T) * N * 3, v, N, LD);
    if (nhc && threadIdx.x < C) A.pv_t[((size_t)rep * T) * C + threadIdx.x] = pv[threadIdx.x];

    bool stale = false;
    if constexpr (KIND < 0) stale = A.code != nullptr;
    if constexpr (KIND < 0) {
        if (stale) force_pairs_stale<DIAG, NT, 1>(A, tpa_log2, q, nullptr, f, nullptr, dth_unused, stale_due(A, 0));
    }
    if (!stale) force_all_pairs<DIAG, NT, KIND, 1>(A, tpa_log2, q, nullptr, f, nullptr, dth_unused, TB, vmax_unused);
    __syncthreads();

    for (int k = 0; k + 1 < T; ++k) {
        const float dt = A.t[k + 1] - A.t[k];
        if constexpr (KIND < 0) {
            // stale lists: the first right-hand side of step k is call 2 k.  It sees the positions of call 2 k - 1, so the
            // cached force is this call's force unless the lists are rebuilt now
            if (stale && k > 0 && stale_due(A, 2ll * k)) {
                force_pairs_stale<DIAG, NT, 1>(A, tpa_log2, q, nullptr, f, nullptr, dth_unused, true);
                __syncthreads();
            }
        }
        // ---- first RHS at y_k (force cached), half kick + drift       sovlers.py:111-118
        float ke = 0.f;
        if (nhc) {
            float part = 0.f;
            MDG_FOR_DOF(e, ia, ca) {
                const float p = v[e] * ms[ia]; part += p * v[e];             // p^2 / m
            }
            ke = 0.5f * block_sum(part, red);
            if (threadIdx.x < C) pb[threadIdx.x] = bath_rhs(A, Qs, pv, ke, threadIdx.x);
        }
        const float pv0 = nhc ? pv[0] : 0.f;
        __syncthreads();
        MDG_FOR_DOF(e, ia, ca) {
            const float m = ms[ia];
            float a;
            if (nhc) { const float p = v[e] * m; a = (f[e] - pv0 * p * iQ0) * __builtin_amdgcn_rcpf(m); }
            else a = f[e];                                   // md.py:145-148 (no 1/m)
            const float h = 0.5f * a * dt;
            vh[e] = h;
            q[e] = q[e] + (v[e] + h) * dt;
        }
        if (nhc && threadIdx.x < C) {
            const float h = 0.5f * pb[threadIdx.x] * dt;
            ph[threadIdx.x] = h;
            pvh[threadIdx.x] = pv[threadIdx.x] + h;
        }
        __syncthreads();
        // ---- second RHS at (v + vh, q1, pv + ph)                      sovlers.py:121-125
        if constexpr (KIND < 0) {
            if (stale) force_pairs_stale<DIAG, NT, 1>(A, tpa_log2, q, nullptr, f, nullptr, dth_unused, stale_due(A, 2ll * k + 1));
        }
        if (!stale) force_all_pairs<DIAG, NT, KIND, 1>(A, tpa_log2, q, nullptr, f, nullptr, dth_unused, TB, vmax_unused);
        float pvh0 = 0.f;
        if (nhc) {
            float part = 0.f;
            MDG_FOR_DOF(e, ia, ca) {
                const float vv = v[e] + vh[e]; const float p = vv * ms[ia]; part += p * vv;
            }
            ke = 0.5f * block_sum(part, red);       // (barriers inside also publish f)
            pvh0 = pvh[0];
            float b1 = 0.f;
            if (threadIdx.x < C) b1 = bath_rhs(A, Qs, pvh, ke, threadIdx.x);
            __syncthreads();
            if (threadIdx.x < C) pv[threadIdx.x] = pv[threadIdx.x] + (ph[threadIdx.x] + 0.5f * b1 * dt);
        } else {
            __syncthreads();
        }
        MDG_FOR_DOF(e, ia, ca) {
            const float m = ms[ia];
            float a;
            if (nhc) { const float p = (v[e] + vh[e]) * m; a = (f[e] - pvh0 * p * iQ0) * __builtin_amdgcn_rcpf(m); }
            else a = f[e];
            v[e] = v[e] + (vh[e] + 0.5f * a * dt);
        }
        __syncthreads();
        store_aos(A.q_t + ((size_t)rep * T + k + 1) * N * 3, q, N, LD);
        store_aos(A.v_t + ((size_t)rep * T + k + 1) * N * 3, v, N, LD);
        if (nhc && threadIdx.x < C) A.pv_t[((size_t)rep * T + k + 1) * C + threadIdx.x] = pv[threadIdx.x];
    }
    if (A.nonfinite) {
        int bad = 0;
        MDG_FOR_DOF(e, ia, ca) bad |= !(isfinite(q[e]) && isfinite(v[e]));
        const int nf = __syncthreads_or(bad) ? 1 : 0;
        int below = 0;
        if constexpr (KIND == KIND_TABLE) below = __syncthreads_or(vmax_unused > 0.f) ? 2 : 0;
        if ((nf | below) && threadIdx.x == 0) A.nonfinite[rep] = nf | below;
    }
}

// ------------------------------------------------------------------------------------ adjoint
// One evaluation of the augmented dynamics at (v,q,pv ; lv,lq,lp):
//   f, dq, th[K] (block-reduced), ke, slv = sum(lv.v)
template <bool DIAG, int NT, int KIND>
__device__ __forceinline__ void aug_eval(const TrajArgs& A, int tpa_log2, bool nhc, const float* q, const float* v,
                                         const float* lv, const float* ms, float* w, float* f,
                                         float* dq, float* red, float (&th)[KMAX], float& ke,
                                         float& slv, const TableRef& TB, float& vmax, bool th_on = true,
                                         int stale_mode = 0 /* 0: no stale lists, 1: stored lists, 2: rebuild */) {
    const int N = A.prm.n_atoms, LD = A.ld;
    MDG_FOR_DOF(e, ia, ca) w[e] = nhc ? lv[e] * __builtin_amdgcn_rcpf(ms[ia]) : lv[e];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KMAX; ++k) th[k] = 0.f;
    bool done = false;
    if constexpr (KIND < 0) {
        if (stale_mode) { force_pairs_stale<DIAG, NT, 2>(A, tpa_log2, q, w, f, dq, th, stale_mode == 2); done = true; }
    }
    if (!done) force_all_pairs<DIAG, NT, KIND, 2>(A, tpa_log2, q, w, f, dq, th, TB, vmax, th_on);
    // one fused block reduction: th[0..K), sum p^2/m, sum lv.v
    float vals[KMAX + 2];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) vals[k] = th[k];
    float p1 = 0.f, p2 = 0.f;
    if (nhc) {
        MDG_FOR_DOF(e, ia, ca) {
            const float m = ms[ia]; const float p = v[e] * m;
            p1 += p * v[e]; p2 += lv[e] * v[e];
        }
    }
    vals[KMAX] = p1; vals[KMAX + 1] = p2;
    block_sum_n<KMAX + 2>(vals, red);
#pragma unroll
    for (int k = 0; k < KMAX; ++k) th[k] = vals[k];
    ke = 0.5f * vals[KMAX]; slv = vals[KMAX + 1];
    __syncthreads();
}

// lam^T d(bath rhs)/d pv_k  + coupling from dv
__device__ __forceinline__ float bath_vjp(const TrajArgs& A, const float* Q, const float* pv, const float* lp,
                                          float slv, int k) {
    const int C = A.prm.n_chains;
    if (k == 0) return -slv / Q[0] - lp[0] * pv[1] / Q[1] + 2.f * pv[0] * lp[1] / Q[0];
    if (k == C - 1) return -lp[C - 2] * pv[C - 2] / Q[C - 1];
    return -lp[k - 1] * pv[k - 1] / Q[k] - lp[k] * pv[k + 1] / Q[k + 1] + 2.f * pv[k] * lp[k + 1] / Q[k];
}

template <bool DIAG, int NT, int KIND>
__global__ __launch_bounds__(1024) void traj_adj_kernel(const TrajArgs A, const int tpa_log2) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N = A.prm.n_atoms, T = A.prm.n_frames, C = A.prm.n_chains, LD = A.ld;
    const bool nhc = A.prm.ensemble == 0;
    const int rep = blockIdx.x, N3 = 3 * N, L3 = 3 * LD;      // N3: frame stride in HBM (AoS), L3: SoA array in LDS
    float* q = smem;        float* v = q + L3;
    float* lv = v + L3;     float* lq = lv + L3;
    float* lvh = lq + L3;   float* lqh = lvh + L3;
    float* w = lqh + L3;    float* f = w + L3;     float* dq = f + L3;
    float* ms = dq + L3;
    float* pv = ms + LD;                      // [C] state
    for (int e = threadIdx.x; e < 28 * LD; e += blockDim.x) smem[e] = 0.f;      // padding columns stay finite
    __syncthreads();
    float* lp = pv + MDG_MAX_CHAINS;          // [C] adjoint
    float* lph = lp + MDG_MAX_CHAINS;         // [C] midpoint adjoint
    float* pb = lph + MDG_MAX_CHAINS;         // [C] scratch: bath rhs
    float* gp = pb + MDG_MAX_CHAINS;          // [C] scratch: bath vjp
    float* Qs = gp + MDG_MAX_CHAINS;          // [C] thermostat masses
    float* red = Qs + MDG_MAX_CHAINS;         // [RED_FLOATS]
#pragma unroll
    for (int c = 0; c < MDG_MAX_CHAINS; ++c)
        if (threadIdx.x == c) Qs[c] = A.prm.Q[c];
    const float Q0 = A.prm.Q[0], iQ0 = 1.f / Q0;
    (void)Q0;   // (per-element divisions by the atom's mass and by Q0 run as v_rcp_f32 / a multiply: 1 ulp)
    float th[KMAX], gth[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) gth[k] = 0.f;
    float vmax = 0.f;
    TableRef TB{nullptr, nullptr, 0.f}, TBacc = TB;
    if constexpr (KIND == KIND_TABLE) {
        const int M = A.terms.t[0].p;
        float2* tab = reinterpret_cast<float2*>(red + RED_FLOATS);
        unsigned long long* g64 = reinterpret_cast<unsigned long long*>(tab + M);      // [2 M] int64 words (8-byte aligned: tab is)
        const float* thp = A.theta + A.terms.t[0].theta_off;
        for (int g = threadIdx.x; g < M; g += blockDim.x) tab[g] = make_float2(thp[2 * g], thp[2 * g + 1]);
        for (int g = threadIdx.x; g < 2 * M; g += blockDim.x) g64[g] = 0ull;
        TB.tab = tab;
        TBacc = TableRef{tab, g64, 0.f};
    }
    const size_t fr = (size_t)rep * T;
    const int tid = threadIdx.x;

    for (int i = tid; i < N; i += blockDim.x) ms[i] = A.mass[i];
    // lam = dL/dy_{T-1}                                            sovlers.py:249
    if (A.g_v) load_soa(lv, A.g_v + (fr + T - 1) * N3, N, LD);          // (else: zero from the fill above)
    if (A.g_q) load_soa(lq, A.g_q + (fr + T - 1) * N3, N, LD);
    if (nhc && tid < C) lp[tid] = A.g_pv ? A.g_pv[(fr + T - 1) * C + tid] : 0.f;

    for (int i = T - 1; i >= 1; --i) {
        const float h = A.t[i] - A.t[i - 1];
        __syncthreads();
        load_soa(q, A.q_t + (fr + i) * N3, N, LD);
        load_soa(v, A.v_t + (fr + i) * N3, N, LD);
        if (nhc && tid < C) pv[tid] = A.pv_t[(fr + i) * C + tid];
        __syncthreads();
        float ke, slv;
        // ---------------- first augmented evaluation at (y_i, lam)
        // (table kind: the parameter term of an interval comes from the midpoint evaluation for NHC,
        //  sovlers.py:160, and from this first one for NVE, :82,101 -- both with total weight h)
        TBacc.gw = 0.5f * h * A.terms.t[0].c;
        // stale lists (topology_update_freq > 1): an interval makes three calls -- the dL/dt evaluation at y_i (sovlers.py:258;
        // result unused, but it advances the counter and may rebuild), the first augmented evaluation at the same
        // positions, the midpoint evaluation -- with running counts c0, c0 + 1, c0 + 2
        int sm1 = 0, sm2 = 0;
        if constexpr (KIND < 0) {
            if (A.code != nullptr) {
                const long long c0 = 3ll * (T - 1 - i);
                sm1 = (stale_due(A, c0) || stale_due(A, c0 + 1)) ? 2 : 1;
                sm2 = stale_due(A, c0 + 2) ? 2 : 1;
            }
        }
        aug_eval<DIAG, NT, KIND>(A, tpa_log2, nhc, q, v, lv, ms, w, f, dq, red, th, ke, slv, nhc ? TB : TBacc, vmax,
                                 /*th_on=*/!nhc, sm1);
        if (nhc) {
            const float pv0 = pv[0], lp0 = lp[0];
            if (tid < C) { pb[tid] = bath_rhs(A, Qs, pv, ke, tid); gp[tid] = bath_vjp(A, Qs, pv, lp, slv, tid); }
            __syncthreads();
            MDG_FOR_DOF(e, ia, ca) {
                const float m = ms[ia], ve = v[e], p = ve * m;
                const float a = (f[e] - pv0 * p * iQ0) * __builtin_amdgcn_rcpf(m);
                const float Gv = -(pv0 * iQ0) * lv[e] + lq[e] + 2.f * m * ve * lp0;
                const float vhalf = 0.5f * (-a) * h;                  // sovlers.py:132
                q[e] = q[e] + (ve + vhalf) * h;                      // :138 forward-time sign (quirk)
                v[e] = ve + vhalf;
                lvh[e] = lv[e] + Gv * 0.5f * h;                      // :141
                lqh[e] = lq[e] + dq[e] * 0.5f * h;                   // :142
            }
            if (tid < C) {
                lph[tid] = lp[tid] + gp[tid] * 0.5f * h;             // :143
            }
            __syncthreads();
            if (tid < C) pv[tid] = pv[tid] + 0.5f * (-pb[tid]) * h;   // :135
            __syncthreads();
            // ---------------- midpoint evaluation                    :147-150
            aug_eval<DIAG, NT, KIND>(A, tpa_log2, nhc, q, v, lvh, ms, w, f, dq, red, th, ke, slv, TBacc, vmax, true, sm2);
            const float pvm0 = pv[0], lpm0 = lph[0];
            if (tid < C) gp[tid] = bath_vjp(A, Qs, pv, lph, slv, tid);
            __syncthreads();
            MDG_FOR_DOF(e, ia, ca) {
                const float m = ms[ia];
                const float Gv = -(pvm0 * iQ0) * lvh[e] + lqh[e] + 2.f * m * v[e] * lpm0;
                float nlv = lv[e] + Gv * h;                          // :156
                float nlq = lq[e] + dq[e] * h;                       // :157
                if (A.g_v) nlv += A.g_v[(fr + i - 1) * N3 + ia * 3 + ca];   // :286
                if (A.g_q) nlq += A.g_q[(fr + i - 1) * N3 + ia * 3 + ca];
                lv[e] = nlv; lq[e] = nlq;
            }
            if (tid < C) {
                float nlp = lp[tid] + gp[tid] * h;                   // :158
                if (A.g_pv) nlp += A.g_pv[(fr + i - 1) * C + tid];
                lp[tid] = nlp;
            }
#pragma unroll
            for (int k = 0; k < KMAX; ++k) gth[k] += th[k] * h;     // :160
        } else {
            // verlet_update backward branch                          sovlers.py:42-101
            MDG_FOR_DOF(e, ia, ca) {
                const float dvv = -f[e];
                const float vhalf = v[e] - 0.5f * dvv * h;           // :49-50
                q[e] = q[e] - vhalf * h;                             // :51-52
                v[e] = vhalf;
                const float dx = dq[e] * h * 0.5f;                   // :71
                const float dvad = (lq[e] + dx) * h;                 // :72
                lvh[e] = lv[e] + dvad;
                lqh[e] = lq[e] + dx;
            }
#pragma unroll
            for (int k = 0; k < KMAX; ++k) gth[k] += (th[k] * 0.5f * h) * 2.f;   // :82,101
            __syncthreads();
            aug_eval<DIAG, NT, KIND>(A, tpa_log2, nhc, q, v, lvh, ms, w, f, dq, red, th, ke, slv, TB, vmax, true, sm2);
            MDG_FOR_DOF(e, ia, ca) {
                float nlv = lvh[e];                                   // lv + dvad
                float nlq = lqh[e] + dq[e] * h * 0.5f;                // :100
                if (A.g_v) nlv += A.g_v[(fr + i - 1) * N3 + ia * 3 + ca];
                if (A.g_q) nlq += A.g_q[(fr + i - 1) * N3 + ia * 3 + ca];
                lv[e] = nlv; lq[e] = nlq;
            }
        }
    }
    __syncthreads();
    store_aos(A.adj_v0 + (size_t)rep * N3, lv, N, LD);
    store_aos(A.adj_q0 + (size_t)rep * N3, lq, N, LD);
    if (nhc && tid < C && A.adj_pv0) A.adj_pv0[(size_t)rep * C + tid] = lp[tid];
    if constexpr (KIND == KIND_TABLE) {
        // table gradient: fixed point -> float; an out-of-range contribution poisons the output (the
        // host re-scales and reports it)
        // (one word can receive a contribution from every directed pair of the accumulating evaluation of every interval)
        const float worst = block_sum(vmax >= fx64_limit((double)(T > 1 ? T - 1 : 1) * (double)N * (double)N) ? 1.f : 0.f, red);
        if (A.adj_theta) {
            const int KT = A.terms.n_theta_total, M2 = 2 * A.terms.t[0].p;
            const double inv = 1.0 / (double)A.terms.t[0].c;
            float* out = A.adj_theta + (size_t)rep * KT + A.terms.t[0].theta_off;
            for (int g = tid; g < M2; g += blockDim.x)
                out[g] = worst > 0.f ? __builtin_inff()
                                     : (float)((double)(long long)TBacc.g64[g] * inv);
        }
        return;
    }
    if (tid == 0 && A.adj_theta) {
        const int KT = A.terms.n_theta_total;
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
            for (int k = 0; k < MDG_MAX_THETA; ++k)
                if (m < A.terms.n_terms && k < A.terms.t[m].n_theta)
                    A.adj_theta[(size_t)rep * KT + A.terms.t[m].theta_off + k] = gth[m * MDG_MAX_THETA + k];
    }
}

#include "traj_ring.hpp"

// ------------------------------------------------------------------------------------ launch
// wave-per-replica kernels (traj_ring.hpp): one unmasked LJ 12-6 term, orthorhombic cell, N <= 128, and either
// asked for (block = 64) or a many-replica launch, where throughput matters and not the latency of one replica
// Tabulated kind on the ring kernels: up to 2 048 nodes.  The adjoint workgroup (RING_TABLE_WAVES = 8 replicas sharing the nodes
// and one set of int64 gradient words) asks for 8 x 3 KB of ring buffers + 24 B per node + 16 B = 73.7 KB at 2 048 nodes: above
// the 64 KB a workgroup gets on older parts, inside gfx950's 160 KB -- ring_form() checks the request against the device's
// own limit and hands a launch that does not fit to the one-workgroup-per-replica kernels.
constexpr int RING_TABLE_MAX_NODES = 2048;
size_t ring_table_adj_lds(int nodes);          // (below, with the launch geometry)
size_t device_lds_per_block() {
    static size_t cached = 0;
    if (!cached) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0)
            v = 64 * 1024;
        cached = (size_t)v;
    }
    return cached;
}

int ring_kind(const MdgPairTerm& t);
bool ring_form(const MdgTrajParams& p, const MdgCell& cell, const MdgTerms& terms) {
    const MdgPairTerm& t = terms.t[0];
    // one unmasked built-in pair form (LJ 12-6 / ExcludedVolume(12) on the even-power polynomial, the others through
    // pair_eval) in an orthorhombic cell
    // (a selection mask -- index_tuple / ex_pairs -- is taken for the LJ family: MDG_RING_LAUNCH)
    // (round 5: the tabulated pair model -- pairMLP + prior stacks -- up to RING_TABLE_MAX_NODES nodes: the nodes and the
    //  replica's gradient planes sit in LDS beside the wave's ring buffers)
    if (t.kind == MDG_PAIR_TABLE) {
        const char* e = getenv("MDG_RING_TABLE");                        // (=0: the one-workgroup-per-replica kernels; A/B)
        return !(e && e[0] == '0') && terms.n_terms == 1 && cell.diag && !t.mask && t.p <= RING_TABLE_MAX_NODES && p.n_atoms <= 128 &&
               ring_table_adj_lds(t.p) <= device_lds_per_block();
    }
    if (terms.n_terms == 2 || terms.n_terms == 3) {
        // round 6: two or three terms of the LJ family with selection masks (species mixtures -- A-A / A-B / B-B index_tuple
        // stacks, interface.py:228-260, scripts/fit_mix.py:101-117) -- one ring sweep per term (ring_force_terms); a term
        // without a mask selects every pair
        bool any_mask = false;
        for (int m = 0; m < terms.n_terms; ++m) {
            const MdgPairTerm& u = terms.t[m];
            if (u.kind != MDG_PAIR_LJ || ring_kind(u) != ring_kind(t)) return false;
            any_mask = any_mask || u.mask != nullptr;
        }
        return cell.diag && p.n_atoms <= 128 && any_mask;
    }
    return terms.n_terms == 1 && cell.diag && (!t.mask || t.kind == MDG_PAIR_LJ) && t.kind >= 0 && t.kind <= MDG_PAIR_YUKAWA &&
           p.n_atoms <= 128;
}
bool use_ring(const MdgTrajParams& p, const MdgCell& cell, const MdgTerms& terms) {
    return ring_form(p, cell, terms) && (p.block == 64 || (p.block == 0 && p.n_rep >= 1024));
}
int ring_kind(const MdgPairTerm& t) {
    if (t.kind == MDG_PAIR_TABLE) return KIND_TABLE;
    return (t.kind == MDG_PAIR_LJ && t.p == 12 && (t.q == 6 || t.c == 0.f)) ? KIND_LJ126 : t.kind;
}
// LDS of the tabulated kind behind a wave's ring buffers: the nodes (forward) + the int64 gradient words (adjoint) + a flag word
size_t ring_table_lds(const MdgTerms& terms, bool adjoint) {
    if (terms.t[0].kind != MDG_PAIR_TABLE) return 0;
    return sizeof(float) * (size_t)(adjoint ? 6 : 2) * terms.t[0].p + 16;
}
// launch of a ring kernel specialised on the pair form
#define MDG_RING_LAUNCH(KERNEL, RDF_, grid, block, lds, st, ...)                                              \
    do {                                                                                                      \
        const bool masked_ = terms->t[0].mask != nullptr;                                                     \
        const int nt_ = terms->n_terms;              /* (ring_form: > 1 only for the LJ family, unfused observable) */ \
        switch (ring_kind(terms->t[0])) {                                                                     \
        case KIND_LJ126:                                                                                      \
            if (nt_ == 2) hipLaunchKernelGGL((KERNEL<false, KIND_LJ126, true, 2>), grid, block, lds, st, __VA_ARGS__);    \
            else if (nt_ == 3) hipLaunchKernelGGL((KERNEL<false, KIND_LJ126, true, 3>), grid, block, lds, st, __VA_ARGS__); \
            else if (masked_) hipLaunchKernelGGL((KERNEL<RDF_, KIND_LJ126, true>), grid, block, lds, st, __VA_ARGS__);    \
            else hipLaunchKernelGGL((KERNEL<RDF_, KIND_LJ126>), grid, block, lds, st, __VA_ARGS__);                       \
            break;                                                                                            \
        case MDG_PAIR_LJ:                                                                                     \
            if (nt_ == 2) hipLaunchKernelGGL((KERNEL<false, MDG_PAIR_LJ, true, 2>), grid, block, lds, st, __VA_ARGS__);   \
            else if (nt_ == 3) hipLaunchKernelGGL((KERNEL<false, MDG_PAIR_LJ, true, 3>), grid, block, lds, st, __VA_ARGS__);  \
            else if (masked_) hipLaunchKernelGGL((KERNEL<RDF_, MDG_PAIR_LJ, true>), grid, block, lds, st, __VA_ARGS__);   \
            else hipLaunchKernelGGL((KERNEL<RDF_, MDG_PAIR_LJ>), grid, block, lds, st, __VA_ARGS__);                      \
            break;                                                                                            \
        case KIND_TABLE: hipLaunchKernelGGL((KERNEL<false, KIND_TABLE>), grid, block, lds, st, __VA_ARGS__); break;     \
        case MDG_PAIR_MORSE: hipLaunchKernelGGL((KERNEL<RDF_, MDG_PAIR_MORSE>), grid, block, lds, st, __VA_ARGS__); break; \
        case MDG_PAIR_BUCK: hipLaunchKernelGGL((KERNEL<RDF_, MDG_PAIR_BUCK>), grid, block, lds, st, __VA_ARGS__); break;  \
        default: hipLaunchKernelGGL((KERNEL<RDF_, MDG_PAIR_YUKAWA>), grid, block, lds, st, __VA_ARGS__); break;          \
        }                                                                                                     \
    } while (0)

constexpr size_t RING_LDS_FWD = sizeof(f32x2) * 3 * 64;      // per wave: the visitors' positions
constexpr size_t RING_LDS_ADJ = sizeof(f32x2) * 6 * 64;      //           ... and adjoint directions
size_t ring_table_adj_lds(int nodes) { return RING_TABLE_WAVES * RING_LDS_ADJ + sizeof(float) * 6 * (size_t)nodes + 16; }
constexpr int RING_RDF_WAVES = 16;                           // waves sharing the fine histogram of the fused RDF
constexpr int RING_RDF_MAX_CELLS = 1088;                     // derivative table <= 17 KB: eight adjoint waves per CU

// the fused RDF observable: fine-grid plan (csrc/rdf.hip) + what fits beside the kernels' own LDS
bool ring_rdf_plan(const MdgTrajParams& p, const MdgCell& cell, const MdgTerms& terms, const MdgRdfFuse* rdf, RdfFinePlan* plan) {
    if (!rdf || !rdf->mu || !use_ring(p, cell, terms)) return false;     // (only where the ring kernels run anyway)
    if (terms.t[0].kind == MDG_PAIR_TABLE || terms.n_terms != 1) return false;   // (the tabulated kind and two-term stacks keep the separate observable kernels)
    if (rdf->frame_stride < 1 || rdf->frame_start < 0 || !(rdf->cutoff > 0.f)) return false;
    const RdfFinePlan P = mdg_rdf_fine_plan(rdf->spacing, rdf->coeff, rdf->nbins);
    if (P.nfine <= 0 || P.ncell <= 0 || P.ncell > RING_RDF_MAX_CELLS) return false;
    // both fine grids must start above zero: the kernels' single range compare then also rejects a zero distance
    if (!(rdf->mu0 - P.reach > 0.f) || !(rdf->mu0 - (float)(P.reach_bins + 1) * rdf->spacing > 0.f)) return false;
    if (sizeof(float) * (size_t)((P.nfine + 1) & ~1LL) + RING_RDF_WAVES * RING_LDS_FWD > 156 * 1024) return false;
    *plan = P;
    return true;
}

RingRdfArgs ring_rdf_args(const MdgRdfFuse& rdf, const RdfFinePlan& P) {
    RingRdfArgs F{};
    F.mu = rdf.mu; F.nbins = rdf.nbins; F.rc = rdf.cutoff;
    F.f_start = rdf.frame_start; F.f_stride = rdf.frame_stride;
    F.reach = P.reach; F.inv_h = 1.0f / P.h; F.nfine = (int)P.nfine; F.reach_bins = P.reach_bins;
    return F;
}

int pick_tpa_log2(int n_atoms, int block) {
    int l = 0;
    while (l < 6 && (n_atoms << (l + 1)) <= block) ++l;
    return l;
}

int pick_block(const MdgTrajParams& p, bool table) {
    if (p.block > 0) return p.block;
    // few replicas: widest workgroup (latency); many replicas: one lane per atom (no cross-lane reduction of
    // the per-atom sums, small workgroups interleave better: 128 vs 256 threads at N = 108 is -11 % adjoint time)
    // (the tabulated kind keeps 256: its gradient scatter into LDS likes two lanes per atom, measured +16 %)
    if (p.n_rep >= 1024 && table) return 256;
    if (p.n_rep >= 1024) {
        const int b = (p.n_atoms + 63) & ~63;
        return b < 128 ? 128 : (b > 1024 ? 1024 : b);
    }
    if (p.n_rep >= 256) return 512;
    return 1024;
}

// specialisation table: single-term kernels with the functional form fixed at compile time
// (orthorhombic cell), everything else through the generic <NT = MDG_MAX_TERMS> kernel.
#define MDG_TRAJ_DISPATCH(KERNEL)                                                                      \
    do {                                                                                               \
        const bool single = terms->n_terms == 1 && diag && !terms->t[0].mask;                          \
        const int kind = terms->t[0].kind;                                                             \
        if (kind == MDG_PAIR_TABLE)                                                                    \
            hipLaunchKernelGGL((KERNEL<true, 1, KIND_TABLE>), grid, dim3(block), lds, st, a, tl);      \
        else if (single && kind == MDG_PAIR_LJ && terms->t[0].p == 12 && (terms->t[0].q == 6 || terms->t[0].c == 0.f)) \
            hipLaunchKernelGGL((KERNEL<true, 1, KIND_LJ126>), grid, dim3(block), lds, st, a, tl);      \
        else if (single && kind == MDG_PAIR_LJ)                                                        \
            hipLaunchKernelGGL((KERNEL<true, 1, MDG_PAIR_LJ>), grid, dim3(block), lds, st, a, tl);     \
        else if (single && kind == MDG_PAIR_MORSE)                                                     \
            hipLaunchKernelGGL((KERNEL<true, 1, MDG_PAIR_MORSE>), grid, dim3(block), lds, st, a, tl);  \
        else if (single && kind == MDG_PAIR_BUCK)                                                      \
            hipLaunchKernelGGL((KERNEL<true, 1, MDG_PAIR_BUCK>), grid, dim3(block), lds, st, a, tl);   \
        else if (single && kind == MDG_PAIR_YUKAWA)                                                    \
            hipLaunchKernelGGL((KERNEL<true, 1, MDG_PAIR_YUKAWA>), grid, dim3(block), lds, st, a, tl); \
        else if (diag)                                                                                 \
            hipLaunchKernelGGL((KERNEL<true, MDG_MAX_TERMS, -1>), grid, dim3(block), lds, st, a, tl);  \
        else                                                                                           \
            hipLaunchKernelGGL((KERNEL<false, MDG_MAX_TERMS, -1>), grid, dim3(block), lds, st, a, tl); \
    } while (0)

int validate(const MdgTrajParams* p, const MdgCell* cell, const MdgTerms* terms) {
    MDG_CHECK_ARG(p && cell && terms, "traj: null descriptor");
    MDG_CHECK_ARG(p->n_rep > 0 && p->n_atoms > 0 && p->n_frames >= 1, "traj: bad sizes R=%d N=%d T=%d",
                  p->n_rep, p->n_atoms, p->n_frames);
    MDG_CHECK_ARG(p->ensemble == 0 || p->ensemble == 1, "traj: ensemble must be 0 (NHC) or 1 (NVE)");
    MDG_CHECK_ARG(p->ensemble == 1 || (p->n_chains >= 2 && p->n_chains <= MDG_MAX_CHAINS),
                  "traj: NoseHooverChain needs 2 <= num_chains <= %d (got %d)", MDG_MAX_CHAINS, p->n_chains);
    MDG_CHECK_ARG(terms->n_terms >= 1 && terms->n_terms <= MDG_MAX_TERMS, "traj: 1..%d pair terms", MDG_MAX_TERMS);
    if (terms->t[0].kind == MDG_PAIR_TABLE) {
        const MdgPairTerm& t = terms->t[0];
        MDG_CHECK_ARG(terms->n_terms == 1 && !t.mask && cell->diag, "traj: a tabulated pair model must be the only "
                      "term, unmasked, in an orthorhombic cell");
        MDG_CHECK_ARG(t.p >= 4 && t.p <= 4096 && t.n_theta == 2 * t.p && t.phi > 0.f && t.c > 0.f,
                      "traj: bad table (nodes %d, n_theta %d, du %g, scale %g)", t.p, t.n_theta, t.phi, t.c);
        return MDG_OK;
    }
    for (int m = 0; m < terms->n_terms; ++m)
        MDG_CHECK_ARG(terms->t[m].kind >= 0 && terms->t[m].kind <= MDG_PAIR_YUKAWA &&
                      terms->t[m].n_theta <= MDG_MAX_THETA, "traj: bad pair term %d", m);
    return MDG_OK;
}

}  // namespace

extern "C" int mdg_traj_fwd_small(const MdgTrajParams* prm, const MdgCell* cell, const MdgTerms* terms,
                                  const float* theta, const float* mass, const float* t_grid,
                                  const float* v0, const float* q0, const float* pv0,
                                  float* v_t, float* q_t, float* pv_t, int32_t* nonfinite, void* stream) {
    int rc = validate(prm, cell, terms);
    if (rc) return rc;
    MDG_CHECK_ARG(mass && t_grid && v0 && q0 && v_t && q_t, "traj_fwd: null buffer");
    MDG_CHECK_ARG(prm->ensemble == 1 || (pv0 && pv_t), "traj_fwd: NHC needs pv0/pv_t");
    TrajArgs a{};
    a.prm = *prm; a.cell = *cell; a.terms = *terms; a.theta = theta; a.mass = mass; a.t = t_grid;
    a.v0 = v0; a.q0 = q0; a.pv0 = pv0; a.v_t = v_t; a.q_t = q_t; a.pv_t = pv_t; a.nonfinite = nonfinite;
    const int N = prm->n_atoms;
    if (use_ring(*prm, *cell, *terms)) {
        MDG_CHECK_ARG(theta || terms->n_theta_total == 0, "traj_fwd: null theta");
        MDG_RING_LAUNCH(traj_fwd_ring_kernel, false, dim3(prm->n_rep), dim3(64), RING_LDS_FWD + ring_table_lds(*terms, false),
                        (hipStream_t)stream, a, RingRdfArgs{});
        MDG_CHECK_LAUNCH("traj_fwd_ring_kernel");
        return MDG_OK;
    }
    const int block = pick_block(*prm, terms->t[0].kind == MDG_PAIR_TABLE);
    const size_t tab = terms->t[0].kind == MDG_PAIR_TABLE ? 2 * (size_t)terms->t[0].p : 0;
    MDG_CHECK_ARG(!tab || theta, "traj_fwd: the table is passed through theta");
    a.ld = N <= 128 ? 128 : (N + 1) & ~1;
    const size_t lds = sizeof(float) * (13 * (size_t)a.ld + 5 * MDG_MAX_CHAINS + RED_FLOATS + tab);
    MDG_CHECK_ARG(lds <= 160 * 1024, "traj_fwd: N=%d does not fit the LDS-resident kernel", N);
    const int tl = pick_tpa_log2(N, block);
    const bool diag = cell->diag != 0;
    dim3 grid(prm->n_rep);
    hipStream_t st = (hipStream_t)stream;
    MDG_TRAJ_DISPATCH(traj_fwd_kernel);
    MDG_CHECK_LAUNCH("traj_fwd_kernel");
    return MDG_OK;
}

extern "C" int mdg_traj_adj_small(const MdgTrajParams* prm, const MdgCell* cell, const MdgTerms* terms,
                                  const float* theta, const float* mass, const float* t_grid,
                                  const float* v_t, const float* q_t, const float* pv_t,
                                  const float* g_v, const float* g_q, const float* g_pv,
                                  float* adj_v0, float* adj_q0, float* adj_pv0, float* adj_theta,
                                  void* stream) {
    int rc = validate(prm, cell, terms);
    if (rc) return rc;
    MDG_CHECK_ARG(mass && t_grid && v_t && q_t && adj_v0 && adj_q0, "traj_adj: null buffer");
    MDG_CHECK_ARG(prm->ensemble == 1 || pv_t, "traj_adj: NHC needs pv_t");
    TrajArgs a{};
    a.prm = *prm; a.cell = *cell; a.terms = *terms; a.theta = theta; a.mass = mass; a.t = t_grid;
    a.v_t = const_cast<float*>(v_t); a.q_t = const_cast<float*>(q_t); a.pv_t = const_cast<float*>(pv_t);
    a.g_v = g_v; a.g_q = g_q; a.g_pv = g_pv;
    a.adj_v0 = adj_v0; a.adj_q0 = adj_q0; a.adj_pv0 = adj_pv0; a.adj_theta = adj_theta;
    const int N = prm->n_atoms;
    if (use_ring(*prm, *cell, *terms)) {
        MDG_CHECK_ARG(theta || terms->n_theta_total == 0, "traj_adj: null theta");
        // (tabulated kind: RING_TABLE_WAVES replicas per workgroup share the nodes and one pair of gradient planes)
        const bool rt = terms->t[0].kind == MDG_PAIR_TABLE;
        const int wpw = rt ? RING_TABLE_WAVES : 1;
        MDG_RING_LAUNCH(traj_adj_ring_kernel, false, dim3((prm->n_rep + wpw - 1) / wpw), dim3(64 * wpw),
                        wpw * RING_LDS_ADJ + ring_table_lds(*terms, true), (hipStream_t)stream, a, RingRdfArgs{});
        MDG_CHECK_LAUNCH("traj_adj_ring_kernel");
        return MDG_OK;
    }
    const int block = pick_block(*prm, terms->t[0].kind == MDG_PAIR_TABLE);
    const size_t tab = terms->t[0].kind == MDG_PAIR_TABLE ? 6 * (size_t)terms->t[0].p : 0;   // nodes + the int64 gradient words
    MDG_CHECK_ARG(!tab || theta, "traj_adj: the table is passed through theta");
    a.ld = N <= 128 ? 128 : (N + 1) & ~1;
    const size_t lds = sizeof(float) * (28 * (size_t)a.ld + 6 * MDG_MAX_CHAINS + RED_FLOATS + tab);
    MDG_CHECK_ARG(lds <= 160 * 1024, "traj_adj: N=%d does not fit the LDS-resident kernel", N);
    const int tl = pick_tpa_log2(N, block);
    const bool diag = cell->diag != 0;
    dim3 grid(prm->n_rep);
    hipStream_t st = (hipStream_t)stream;
    MDG_TRAJ_DISPATCH(traj_adj_kernel);
    MDG_CHECK_LAUNCH("traj_adj_kernel");
    return MDG_OK;
}

// ------------------------------------------------------------------------------------ stale neighbour lists
// mdg_traj_fwd_small / mdg_traj_adj_small for integrators with topology_update_freq > 1 (torchmd/md.py:200-204): see
// TrajArgs::code.  Always the generic multi-term kernels (built-in pair forms, masks, any cell); a tabulated pair model
// is not taken.
extern "C" int64_t mdg_traj_stale_words(int n_rep, int n_atoms) { return (int64_t)n_rep * n_atoms * n_atoms; }

extern "C" int mdg_traj_fwd_small_stale(const MdgTrajParams* prm, const MdgCell* cell, const MdgTerms* terms,
                                        const float* theta, const float* mass, const float* t_grid,
                                        const float* v0, const float* q0, const float* pv0,
                                        float* v_t, float* q_t, float* pv_t, int32_t* nonfinite,
                                        int freq, int64_t count0, uint16_t* code, void* stream) {
    int rc = validate(prm, cell, terms);
    if (rc) return rc;
    MDG_CHECK_ARG(mass && t_grid && v0 && q0 && v_t && q_t, "traj_fwd_stale: null buffer");
    MDG_CHECK_ARG(prm->ensemble == 1 || (pv0 && pv_t), "traj_fwd_stale: NHC needs pv0/pv_t");
    MDG_CHECK_ARG(freq >= 1 && count0 >= 0 && code, "traj_fwd_stale: bad frequency / counter / list buffer");
    MDG_CHECK_ARG(terms->t[0].kind != MDG_PAIR_TABLE, "traj_fwd_stale: a tabulated pair model is not supported");
    TrajArgs a{};
    a.prm = *prm; a.cell = *cell; a.terms = *terms; a.theta = theta; a.mass = mass; a.t = t_grid;
    a.v0 = v0; a.q0 = q0; a.pv0 = pv0; a.v_t = v_t; a.q_t = q_t; a.pv_t = pv_t; a.nonfinite = nonfinite;
    a.code = code; a.freq = freq; a.count0 = count0;
    const int N = prm->n_atoms;
    const int block = pick_block(*prm, false);
    a.ld = N <= 128 ? 128 : (N + 1) & ~1;
    const size_t lds = sizeof(float) * (13 * (size_t)a.ld + 5 * MDG_MAX_CHAINS + RED_FLOATS);
    MDG_CHECK_ARG(lds <= 160 * 1024, "traj_fwd_stale: N=%d does not fit the LDS-resident kernel", N);
    const int tl = pick_tpa_log2(N, block);
    dim3 grid(prm->n_rep);
    hipStream_t st = (hipStream_t)stream;
    if (cell->diag) hipLaunchKernelGGL((traj_fwd_kernel<true, MDG_MAX_TERMS, -1>), grid, dim3(block), lds, st, a, tl);
    else hipLaunchKernelGGL((traj_fwd_kernel<false, MDG_MAX_TERMS, -1>), grid, dim3(block), lds, st, a, tl);
    MDG_CHECK_LAUNCH("traj_fwd_kernel (stale lists)");
    return MDG_OK;
}

extern "C" int mdg_traj_adj_small_stale(const MdgTrajParams* prm, const MdgCell* cell, const MdgTerms* terms,
                                        const float* theta, const float* mass, const float* t_grid,
                                        const float* v_t, const float* q_t, const float* pv_t,
                                        const float* g_v, const float* g_q, const float* g_pv,
                                        float* adj_v0, float* adj_q0, float* adj_pv0, float* adj_theta,
                                        int freq, int64_t count0, uint16_t* code, void* stream) {
    int rc = validate(prm, cell, terms);
    if (rc) return rc;
    MDG_CHECK_ARG(mass && t_grid && v_t && q_t && adj_v0 && adj_q0, "traj_adj_stale: null buffer");
    MDG_CHECK_ARG(prm->ensemble == 1 || pv_t, "traj_adj_stale: NHC needs pv_t");
    MDG_CHECK_ARG(freq >= 1 && count0 >= 0 && code, "traj_adj_stale: bad frequency / counter / list buffer");
    MDG_CHECK_ARG(terms->t[0].kind != MDG_PAIR_TABLE, "traj_adj_stale: a tabulated pair model is not supported");
    TrajArgs a{};
    a.prm = *prm; a.cell = *cell; a.terms = *terms; a.theta = theta; a.mass = mass; a.t = t_grid;
    a.v_t = const_cast<float*>(v_t); a.q_t = const_cast<float*>(q_t); a.pv_t = const_cast<float*>(pv_t);
    a.g_v = g_v; a.g_q = g_q; a.g_pv = g_pv;
    a.adj_v0 = adj_v0; a.adj_q0 = adj_q0; a.adj_pv0 = adj_pv0; a.adj_theta = adj_theta;
    a.code = code; a.freq = freq; a.count0 = count0;
    const int N = prm->n_atoms;
    const int block = pick_block(*prm, false);
    a.ld = N <= 128 ? 128 : (N + 1) & ~1;
    const size_t lds = sizeof(float) * (28 * (size_t)a.ld + 6 * MDG_MAX_CHAINS + RED_FLOATS);
    MDG_CHECK_ARG(lds <= 160 * 1024, "traj_adj_stale: N=%d does not fit the LDS-resident kernel", N);
    const int tl = pick_tpa_log2(N, block);
    dim3 grid(prm->n_rep);
    hipStream_t st = (hipStream_t)stream;
    if (cell->diag) hipLaunchKernelGGL((traj_adj_kernel<true, MDG_MAX_TERMS, -1>), grid, dim3(block), lds, st, a, tl);
    else hipLaunchKernelGGL((traj_adj_kernel<false, MDG_MAX_TERMS, -1>), grid, dim3(block), lds, st, a, tl);
    MDG_CHECK_LAUNCH("traj_adj_kernel (stale lists)");
    return MDG_OK;
}

// ------------------------------------------------------------------------------------ fused RDF observable
extern "C" int mdg_traj_rdf_supported(const MdgTrajParams* prm, const MdgCell* cell, const MdgTerms* terms,
                                      const MdgRdfFuse* rdf) {
    RdfFinePlan P;
    return prm && cell && terms && validate(prm, cell, terms) == MDG_OK && ring_rdf_plan(*prm, *cell, *terms, rdf, &P);
}

extern "C" int mdg_traj_fwd_small_rdf(const MdgTrajParams* prm, const MdgCell* cell, const MdgTerms* terms,
                                      const float* theta, const float* mass, const float* t_grid,
                                      const float* v0, const float* q0, const float* pv0,
                                      float* v_t, float* q_t, float* pv_t, int32_t* nonfinite,
                                      const MdgRdfFuse* rdf, float* raw, void* stream) {
    int rc = validate(prm, cell, terms);
    if (rc) return rc;
    MDG_CHECK_ARG((theta || terms->n_theta_total == 0) && mass && t_grid && v0 && q0 && v_t && q_t && raw, "traj_fwd_rdf: null buffer");
    MDG_CHECK_ARG(prm->ensemble == 1 || (pv0 && pv_t), "traj_fwd_rdf: NHC needs pv0/pv_t");
    RdfFinePlan P;
    MDG_CHECK_ARG(ring_rdf_plan(*prm, *cell, *terms, rdf, &P), "traj_fwd_rdf: not available for this system / observable "
                  "(see mdg_traj_rdf_supported)");
    TrajArgs a{};
    a.prm = *prm; a.cell = *cell; a.terms = *terms; a.theta = theta; a.mass = mass; a.t = t_grid;
    a.v0 = v0; a.q0 = q0; a.pv0 = pv0; a.v_t = v_t; a.q_t = q_t; a.pv_t = pv_t; a.nonfinite = nonfinite;
    hipStream_t st = (hipStream_t)stream;
    uint32_t* ghist = nullptr;
    MDG_HIP(hipMallocAsync((void**)&ghist, sizeof(uint32_t) * (size_t)P.nfine, st));
    MDG_HIP(hipMemsetAsync(ghist, 0, sizeof(uint32_t) * (size_t)P.nfine, st));
    RingRdfArgs F = ring_rdf_args(*rdf, P);
    F.ghist = ghist;
    int grid = (prm->n_rep + RING_RDF_WAVES - 1) / RING_RDF_WAVES;
    if (grid > 256) grid = 256;                                   // one resident workgroup (16 waves) per CU
    const size_t lds = sizeof(float) * (size_t)((P.nfine + 1) & ~1LL) + RING_RDF_WAVES * RING_LDS_FWD;
    MDG_RING_LAUNCH(traj_fwd_ring_kernel, true, dim3(grid), dim3(64 * RING_RDF_WAVES), lds, st, a, F);
    rc = mdg_rdf_fine_finish(ghist, P, rdf->mu, rdf->nbins, raw, st);
    (void)hipFreeAsync(ghist, st);
    if (rc) return rc;
    MDG_CHECK_LAUNCH("traj_fwd_ring_kernel<rdf>");
    return MDG_OK;
}

extern "C" int mdg_traj_adj_small_rdf(const MdgTrajParams* prm, const MdgCell* cell, const MdgTerms* terms,
                                      const float* theta, const float* mass, const float* t_grid,
                                      const float* v_t, const float* q_t, const float* pv_t,
                                      const float* g_v, const float* g_q, const float* g_pv,
                                      float* adj_v0, float* adj_q0, float* adj_pv0, float* adj_theta,
                                      const MdgRdfFuse* rdf, const float* g_raw, void* stream) {
    int rc = validate(prm, cell, terms);
    if (rc) return rc;
    MDG_CHECK_ARG((theta || terms->n_theta_total == 0) && mass && t_grid && v_t && q_t && adj_v0 && adj_q0 && g_raw, "traj_adj_rdf: null buffer");
    MDG_CHECK_ARG(prm->ensemble == 1 || pv_t, "traj_adj_rdf: NHC needs pv_t");
    RdfFinePlan P;
    MDG_CHECK_ARG(ring_rdf_plan(*prm, *cell, *terms, rdf, &P), "traj_adj_rdf: not available for this system / observable "
                  "(see mdg_traj_rdf_supported)");
    TrajArgs a{};
    a.prm = *prm; a.cell = *cell; a.terms = *terms; a.theta = theta; a.mass = mass; a.t = t_grid;
    a.v_t = const_cast<float*>(v_t); a.q_t = const_cast<float*>(q_t); a.pv_t = const_cast<float*>(pv_t);
    a.g_v = g_v; a.g_q = g_q; a.g_pv = g_pv;
    a.adj_v0 = adj_v0; a.adj_q0 = adj_q0; a.adj_pv0 = adj_pv0; a.adj_theta = adj_theta;
    hipStream_t st = (hipStream_t)stream;
    float4* tab = nullptr;                                        // (stream-ordered scratch: no state, re-entrant)
    MDG_HIP(hipMallocAsync((void**)&tab, sizeof(float4) * (size_t)P.ncell, st));
    rc = mdg_rdf_bwd_table_u(rdf->mu, rdf->coeff, rdf->nbins, g_raw, P, tab, st);
    if (rc == MDG_OK) {
        RingRdfArgs F = ring_rdf_args(*rdf, P);
        F.tab = tab;
        MDG_RING_LAUNCH(traj_adj_ring_kernel, true, dim3(prm->n_rep), dim3(64), sizeof(float4) * (size_t)P.ncell + RING_LDS_ADJ, st, a, F);
    }
    (void)hipFreeAsync(tab, st);
    if (rc) return rc;
    MDG_CHECK_LAUNCH("traj_adj_ring_kernel<rdf>");
    return MDG_OK;
}
