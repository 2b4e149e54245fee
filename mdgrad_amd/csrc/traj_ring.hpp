// Wave-per-replica, register-resident variant of the fused trajectory kernels (included by traj_small.hip
// inside its anonymous namespace; TrajArgs and the bath formulas are the ones defined there).
//
// Many-replica launches of a small system (N <= 128, one unmasked LJ 12-6 term, orthorhombic cell) are bound by
// VALU issue, and the all-pairs sweep of traj_*_kernel evaluates every pair twice (once from each end) and
// fetches the j-side operands from LDS.  Here ONE wave integrates one replica with the whole state in VGPRs:
//
//   * lane l owns atoms (2l, 2l+1) as the two halves of packed f32x2 registers -- every update of the
//     integrator and of the adjoint is v_pk_*_f32 on both atoms, no LDS, no barrier;
//   * the force sweep is a systolic ring with Newton's third law over the nl = ceil(N/2) lanes that own atoms:
//     at step k lane l meets the atoms of lane (l - k) mod nl and evaluates the four pairs between its two atoms
//     and the two visitors as two packed operations ("straight" (i0,j0),(i1,j1) and "crossed" (i0,j1),(i1,j0)
//     -- the crossing is an op_sel modifier, not an instruction), adding +F to its own accumulators and -F to
//     the visitors' accumulators, which travel with them from lane to lane (ds_bpermute_b32: the LDS crossbar,
//     no VALU slot).  The visitors' positions and adjoint direction w are read from a 3 KB LDS copy written once
//     per evaluation (conflict-free ds_read_b64).  Steps 1..(nl-1)/2 cover every pair of lanes once; step 0 (the
//     pair inside a lane) and, for even nl, the antipodal step nl/2 (lanes that meet from both sides) are
//     evaluated in both directions without the travelling update.  Afterwards one more bpermute brings the
//     travelling accumulators home.  N = 108: 55 packed pair operations per evaluation instead of 2 x 54.
//     (A 64-lane ring rotated by DPP wave_ror:1 -- full rate, tools/micro/dpp_rot.hip -- spends 65 operations
//     and 24 VALU moves per step: measured slower.)
//   * sums over atoms (kinetic energy, lam.v, the two LJ parameter sums) are wave reductions; the Nose-Hoover
//     chain lives one entry per lane (neighbours through DPP row shifts).
//
// Summation order differs from traj_*_kernel (tolerance-level differences, tests/test_gpu_pins.py); the pair
// set is identical: the same un-contracted |D|^2 < rc^2 test on the same minimum-image components.
// Semantics: torchmd/sovlers.py:110-127 / :25-40 (forward), :211-293 with :129-164 / :42-101 (adjoint),
// torchmd/md.py:210-240 / :133-150 (right-hand sides), torchmd/topology.py:59-67 (pair set).

// ------------------------------------------------------------------------------------------------ lane moves
__device__ __forceinline__ float ring_move(float v, int src4) {      // value of lane src4 / 4
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src4, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ f32x2 ring_move(f32x2 v, int src4) { return f32x2{ring_move(v.x, src4), ring_move(v.y, src4)}; }
__device__ __forceinline__ float row_up(float v) {              // lane l <- lane l+1 within a row of 16, else 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_down(float v) {            // lane l <- lane l-1 within a row of 16, else 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane0(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ f32x2 rcp2(f32x2 v) { return f32x2{__builtin_amdgcn_rcpf(v.x), __builtin_amdgcn_rcpf(v.y)}; }
__device__ __forceinline__ float hsum(f32x2 v) { return v.x + v.y; }

struct Vec3x2 { f32x2 x, y, z; };
__device__ __forceinline__ Vec3x2 vzero() { const f32x2 z = {0.f, 0.f}; return Vec3x2{z, z, z}; }
__device__ __forceinline__ Vec3x2 ring_move(const Vec3x2& a, int s) { return Vec3x2{ring_move(a.x, s), ring_move(a.y, s), ring_move(a.z, s)}; }

// ------------------------------------------------------------------------------------------------ pair sweep
struct RingLJ {
    float sig2, rc2, m1a, m1b, ka, kb, tsa, tsb, tea, teb;      // LJ 12-6 polynomial (KIND_LJ126)
    float ivx, ivy, ivz, hx, hy, hz;
    TermConst t0;                                               // any other single-term form goes through pair_eval
};

__device__ __forceinline__ RingLJ ring_constants(const TrajArgs& A, int m = 0) {
    const TermConst t0 = term_prepare(A.terms.t[m], A.theta);
    const float e4 = 4.f * t0.k1, cq = t0.c;
    RingLJ K;
    K.sig2 = t0.k0 * t0.k0; K.rc2 = t0.rc2;
    K.m1a = 6.f * e4 * cq; K.m1b = 12.f * e4;                          // phi'/r  = (m1a s6 - m1b s12) / d2
    K.ka = (42.f + 6.f) * e4 * cq; K.kb = (156.f + 12.f) * e4;         // phi'' - phi'/r = (kb s12 - ka s6) / d2
    K.tsa = 18.f * e4 * t0.k2 * cq; K.tsb = 72.f * e4 * t0.k2;         // d(w.F)/dsig from (S6, S12)
    K.tea = 12.f * cq; K.teb = 24.f;                                   // d(w.F)/deps
    K.ivx = A.cell.inv[0]; K.ivy = A.cell.inv[4]; K.ivz = A.cell.inv[8];
    K.hx = A.cell.h[0]; K.hy = A.cell.h[4]; K.hz = A.cell.h[8];
    K.t0 = t0;
    return K;
}

// Fused RDF observable (torchmd/observable.py:62-76 on the frames of this trajectory, see csrc/rdf.hip for the two
// formulations it shares): the sweep already holds |D|^2 of every pair of a frame, so
//   RDF = 1 (forward)  counts each pair on the fine integer grid of rdf_fwd_fine_kernel (one ds_add_u32 into the
//                      workgroup's LDS histogram) instead of re-reading the frame in a second kernel;
//   RDF = 2 (adjoint)  evaluates dL/dd from the cubic-Hermite cell table of rdf_bwd_fine_kernel and accumulates the
//                      frame gradient dL/dq_t that the adjoint would otherwise load from HBM.
struct RingRdf {
    uint32_t* hist; float inv_h, tlo, fmax;                 // RDF = 1: fine histogram [nfine], t = d inv_h + tlo < fmax
    const float4* tab; float ulo, inv_hu, tmax;             // RDF = 2: cell cubics of (dL/dd)/d in u = d^2, t = (d2 - ulo) inv_hu < tmax
                                                            // (fmax / tmax: end of the grid or the observable's cutoff)
    // KIND_TABLE (round 5; the tabulated pair model of traj_small.hip force_table_packed -- pairMLP + prior stacks,
    // scripts/fit_rdf_pair.py:355-368 -- register-resident): the nodes (c1_g, du dc1/du_g) of c1(u) = phi'(r)/r in LDS; the
    // fixed-point words of this workgroup's table gradient (fx64, common.hpp: one int64 per entry, integer LDS atomics:
    // order-independent) and the weight of the current evaluation's contributions (0: the evaluation does not accumulate,
    // sovlers.py:160 / :82,101); tlim: the largest single contribution the words can take without any sum leaving int64;
    // tflag: LDS word, bit 1 = a live pair below the first node (forward), bit 2 = a contribution at or beyond tlim (adjoint)
    const float2* ttab; unsigned long long* tg64; int32_t* tflag; float tgw, tu0, tinv_du, ttmax, tlim;
};

// A node's value and slope entries (idx even, idx + 1): one 64-bit LDS atomic each -- as many atomic instructions as the
// two packed int32 planes of round 5 took for the pair, and no carry between neighbours to go wrong.
__device__ __forceinline__ void ring_table_scatter2(const RingRdf& X, int idx, float va, float vb) {
    if (!(fmaxf(fabsf(va), fabsf(vb)) < X.tlim)) *X.tflag = 4;            // out of range (or not a number): the host re-scales
    atomicAdd(X.tg64 + idx, fx64(va));
    atomicAdd(X.tg64 + idx + 1, fx64(vb));
}

// One packed pair operation: lane atoms (i0, i1) against visitors (j0, j1) [CROSS: (j1, j0)].
// v0 / v1: both atoms of the pair in .x / .y exist.  JSIDE: also update the visitors' accumulators.
// LEVEL 0: geometry only (RDF gradient of a frame the adjoint does not evaluate forces at).
// r0 / r1: the pair in .x / .y feeds the RDF (exists and, for RDF = 1, is the one copy of a pair met twice).
template <int LEVEL, bool NEAR, bool CROSS, bool JSIDE, int RDF, int KIND>
__device__ __forceinline__ void ring_pair(const RingLJ& K, const RingRdf& X, const Vec3x2& qi, const Vec3x2& wi,
                                          const Vec3x2& qj, const Vec3x2& wj, bool v0, bool v1, bool r0, bool r1,
                                          Vec3x2& fi, Vec3x2& gi, Vec3x2& fj, Vec3x2& gj, f32x2 (&TH)[MDG_MAX_THETA],
                                          Vec3x2& ri, Vec3x2& rj) {
    f32x2 dx = (CROSS ? qj.x.yx : qj.x) - qi.x, dy = (CROSS ? qj.y.yx : qj.y) - qi.y,
          dz = (CROSS ? qj.z.yx : qj.z) - qi.z;                                       // D = x_j - x_i
    if constexpr (NEAR) {
        dx = min_image_diag2_near(dx, K.ivx, K.hx); dy = min_image_diag2_near(dy, K.ivy, K.hy);
        dz = min_image_diag2_near(dz, K.ivz, K.hz);
    } else {
        dx = min_image_diag2(dx, K.ivx, K.hx); dy = min_image_diag2(dy, K.ivy, K.hy); dz = min_image_diag2(dz, K.ivz, K.hz);
    }
    const f32x2 d2 = norm2_ref2(dx, dy, dz);
    // Range tests of the fused observable: for a float t, `bits(t) < bits(tmax)` as unsigned integers is exactly
    // 0 <= t < tmax (a negative t has the sign bit set, a NaN lies above every finite value) -- one compare per
    // pair.  The host folds the observable's own cutoff into tmax / fmax, and the grids start above zero, so
    // this accepts precisely the pairs of topology.py:67 that can contribute.
    if constexpr (RDF == 1) {
        // rdf_fine_frame: t = sqrt(d2) / h - lo / h
        const float ta = fmaf(__builtin_amdgcn_sqrtf(d2.x), X.inv_h, X.tlo), tb = fmaf(__builtin_amdgcn_sqrtf(d2.y), X.inv_h, X.tlo);
        const uint32_t lim = __float_as_uint(X.fmax);
        if (r0 && __float_as_uint(ta) < lim) atomicAdd(&X.hist[(int)ta], 1u);
        if (r1 && __float_as_uint(tb) < lim) atomicAdd(&X.hist[(int)tb], 1u);
    }
    if constexpr (RDF == 2) {
        // rdf_bwd_fine_kernel's gradient, d(dist)/dx_j = +D/d, d(dist)/dx_i = -D/d, from the table of (dL/dd)/d over
        // u = d^2 (rdf_bwd_table_u_kernel): no square root, no division; a rejected pair adds +-0
        // (d2 = 0: t = -ulo inv_hu < 0 -> rejected: the grid starts above zero)
        f32x2 t = (d2 - X.ulo) * X.inv_hu;
        const uint32_t lim = __float_as_uint(X.tmax);
        const bool oka = r0 && __float_as_uint(t.x) < lim, okb = r1 && __float_as_uint(t.y) < lim;
        t = f32x2{oka ? t.x : 0.f, okb ? t.y : 0.f};
        const int gA = (int)t.x, gB = (int)t.y;
        const f32x2 fr = t - f32x2{(float)gA, (float)gB};
        const float4 ca = X.tab[gA], cb = X.tab[gB];
        const float sdA = fmaf(fr.x, fmaf(fr.x, fmaf(fr.x, ca.w, ca.z), ca.y), ca.x);
        const float sdB = fmaf(fr.y, fmaf(fr.y, fmaf(fr.y, cb.w, cb.z), cb.y), cb.x);
        const f32x2 cw = {oka ? sdA : 0.f, okb ? sdB : 0.f};
        const f32x2 cx = cw * dx, cy = cw * dy, cz = cw * dz;
        ri.x -= cx; ri.y -= cy; ri.z -= cz;
        if constexpr (JSIDE) { rj.x += CROSS ? cx.yx : cx; rj.y += CROSS ? cy.yx : cy; rj.z += CROSS ? cz.yx : cz; }
    }
    if constexpr (LEVEL >= 1) {
        const bool ok0 = v0 && (d2.x != 0.f) && (d2.x < K.rc2);                       // topology.py:67
        const bool ok1 = v1 && (d2.y != 0.f) && (d2.y < K.rc2);
        // per pair: c1 = phi'/r, kk = (phi'' - phi'/r)/r^2 and the parameter factors tk (dth_k += tk (w.D) per directed
        // pair) -- from the even-power polynomial for LJ 12-6, from pair_eval for every other form
        f32x2 c1, kk, tk[MDG_MAX_THETA];
        constexpr int NTH = KIND == KIND_LJ126 ? 2 : (KIND == KIND_TABLE ? 0 : kind_ntheta(KIND));
        f32x2 h00 = {0.f, 0.f}, h10 = h00, h01 = h00, h11 = h00;                      // (KIND_TABLE: Hermite basis of the pair's cell)
        int g0 = 0, g1 = 0;
        if constexpr (KIND == KIND_TABLE) {
            // cubic Hermite in u = d^2 on the LDS-resident nodes: c1 = phi'/r, and (phi'' - phi'/r)/r^2 = 2 dc1/du from the
            // derivative of the SAME interpolant (force and Hessian.w stay consistent) -- force_table_packed's arithmetic
            const f32x2 sel = {ok0 ? 1.f : 0.f, ok1 ? 1.f : 0.f};
            if constexpr (LEVEL == 1) {
                if ((ok0 && d2.x < X.tu0) || (ok1 && d2.y < X.tu0)) *X.tflag = 2;  // below the first node: the host raises
            }
            f32x2 tt = (d2 - X.tu0) * X.tinv_du;
            tt.x = fminf(fmaxf(ok0 ? tt.x : 0.f, 0.f), X.ttmax); tt.y = fminf(fmaxf(ok1 ? tt.y : 0.f, 0.f), X.ttmax);
            g0 = (int)tt.x; g1 = (int)tt.y;
            const f32x2 fr = {tt.x - (float)g0, tt.y - (float)g1};
            const float2 a0 = X.ttab[g0], b0 = X.ttab[g0 + 1], a1 = X.ttab[g1], b1 = X.ttab[g1 + 1];
            const f32x2 v0 = {a0.x, a1.x}, s0 = {a0.y, a1.y}, v1 = {b0.x, b1.x}, s1 = {b0.y, b1.y};
            const f32x2 om = 1.f - fr, fr2 = fr * fr, om2 = om * om;
            h00 = (1.f + 2.f * fr) * om2; h10 = fr * om2; h01 = fr2 * (3.f - 2.f * fr); h11 = fr2 * (fr - 1.f);
            c1 = (h00 * v0 + h10 * s0 + h01 * v1 + h11 * s1) * sel;
            if constexpr (LEVEL >= 2) {
                const f32x2 e00 = 6.f * fr * (fr - 1.f), e10 = (3.f * fr - 4.f) * fr + 1.f, e11 = (3.f * fr - 2.f) * fr;
                kk = (2.f * (e00 * (v0 - v1) + e10 * s0 + e11 * s1)) * (X.tinv_du * sel);
            }
        } else if constexpr (KIND == KIND_LJ126) {
            // 1/d2 selected to 0 for a rejected pair: s6, s12 and everything below are then exactly zero
            const f32x2 i2 = {ok0 ? __builtin_amdgcn_rcpf(d2.x) : 0.f, ok1 ? __builtin_amdgcn_rcpf(d2.y) : 0.f};
            const f32x2 s2 = K.sig2 * i2;
            const f32x2 s6 = s2 * s2 * s2;
            const f32x2 s12 = s6 * s6;
            c1 = (K.m1a * s6 - K.m1b * s12) * i2;
            if constexpr (LEVEL >= 2) {
                kk = (K.kb * s12 - K.ka * s6) * (i2 * i2);
                tk[0] = s6 * i2; tk[1] = s12 * i2;       // (S6, S12: both parameter gradients are linear in these sums)
            }
        } else {
            // branch-free: a rejected pair is evaluated at the cutoff and multiplied by zero
            PairOut o0, o1;
            float r0, ir0, r1, ir1;
            pair_eval<LEVEL, KIND>(K.t0, ok0 ? d2.x : K.rc2, r0, ir0, o0);
            pair_eval<LEVEL, KIND>(K.t0, ok1 ? d2.y : K.rc2, r1, ir1, o1);
            c1 = f32x2{ok0 ? o0.du * ir0 : 0.f, ok1 ? o1.du * ir1 : 0.f};
            if constexpr (LEVEL >= 2) {
                kk = f32x2{ok0 ? (o0.d2u - o0.du * ir0) * (ir0 * ir0) : 0.f, ok1 ? (o1.d2u - o1.du * ir1) * (ir1 * ir1) : 0.f};
#pragma unroll
                for (int k = 0; k < NTH; ++k)
                    tk[k] = f32x2{ok0 ? 0.5f * o0.ddu_dth[k] * ir0 : 0.f, ok1 ? 0.5f * o1.ddu_dth[k] * ir1 : 0.f};
            }
        }
        fi.x += c1 * dx; fi.y += c1 * dy; fi.z += c1 * dz;                            // F_i += (phi'/r) D
        if constexpr (JSIDE) {
            if constexpr (CROSS) {
                fj.x = __builtin_elementwise_fma(-c1.yx, dx.yx, fj.x); fj.y = __builtin_elementwise_fma(-c1.yx, dy.yx, fj.y);
                fj.z = __builtin_elementwise_fma(-c1.yx, dz.yx, fj.z);
            } else {
                fj.x -= c1 * dx; fj.y -= c1 * dy; fj.z -= c1 * dz;
            }
        }
        if constexpr (LEVEL >= 2) {
            const f32x2 ax = wi.x - (CROSS ? wj.x.yx : wj.x), ay = wi.y - (CROSS ? wj.y.yx : wj.y),
                        az = wi.z - (CROSS ? wj.z.yx : wj.z);
            const f32x2 b = dx * ax + dy * ay + dz * az;                              // w_ij . D
            const f32x2 k2 = kk * b;                                                  // (phi'' - phi'/r)(w.D)/d2
            const f32x2 tx = __builtin_elementwise_fma(k2, dx, c1 * ax), ty = __builtin_elementwise_fma(k2, dy, c1 * ay),
                        tz = __builtin_elementwise_fma(k2, dz, c1 * az);              // -(H w) contribution
            gi.x += tx; gi.y += ty; gi.z += tz;
            if constexpr (JSIDE) {
                gj.x -= CROSS ? tx.yx : tx; gj.y -= CROSS ? ty.yx : ty; gj.z -= CROSS ? tz.yx : tz;
            }
#pragma unroll
            for (int k = 0; k < NTH; ++k) TH[k] += tk[k] * b;
            if constexpr (KIND == KIND_TABLE) {
                // table gradient: d(w.F)/dnode += 1/2 (D.w_ij) basis per DIRECTED pair (tgw carries 1/2 h 2^S); a ring step
                // meets an undirected pair once and stands for both directions
                if (X.tgw != 0.f) {
                    const f32x2 x = (X.tgw * (JSIDE ? 2.f : 1.f)) * b;
                    if (ok0) {
                        ring_table_scatter2(X, 2 * g0, x.x * h00.x, x.x * h10.x);
                        ring_table_scatter2(X, 2 * g0 + 2, x.x * h01.x, x.x * h11.x);
                    }
                    if (ok1) {
                        ring_table_scatter2(X, 2 * g1, x.y * h00.y, x.y * h10.y);
                        ring_table_scatter2(X, 2 * g1 + 2, x.y * h01.y, x.y * h11.y);
                    }
                }
            }
        }
    }
}

// The lanes of one wave exchange data through LDS: order its accesses (DS instructions of a wave execute in issue
// order, so a compiler-level fence is all that is needed; no s_barrier -- the waves of a workgroup are independent)
__device__ __forceinline__ void ring_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// All pair terms of one replica.  Outputs: f (force), g (= dq of the augmented dynamics, already negated),
// th[k] = this lane's part of the parameter sums over DIRECTED pairs (LEVEL 2; LJ 12-6: of s6 (w.D)/d2 and
// s12 (w.D)/d2, other forms: of 1/2 d2phi/(dr dtheta_k) (w.D)/r), rq = dL/dq of the fused RDF for this frame (RDF = 2).
// The ring has nl = ceil(N/2) lanes (the lanes that own atoms).  The visitors' positions and w do not move at all:
// they sit in LDS ([6][64] f32x2, written once per evaluation) and lane l reads entry (l - k) mod nl at step k;
// only the visitors' accumulators travel, through ds_bpermute_b32 (the LDS crossbar: no VALU slot, no memory).
// MASK: the term carries a selection mask (index_tuple / ex_pairs, topology.py:37-53).  mk[0..3] / mk[4..7] = the 128-bit
// rows of the lane's two atoms (bit j: the pair with atom j is selected); the visitors of lane m are atoms 2m, 2m + 1, i.e.
// bits 2 (m & 15), + 1 of word m >> 4 -- a select chain over four registers and a shift per ring step.  The fused
// observable's flags stay unmasked (its own selection is the host's concern: only unmasked observables are fused).
struct RingMask { uint32_t w[8]; };
__device__ __forceinline__ uint32_t ring_mask_word(const uint32_t* w4, int q) {
    return q == 0 ? w4[0] : (q == 1 ? w4[1] : (q == 2 ? w4[2] : w4[3]));
}
__device__ __forceinline__ RingMask ring_mask_load(const uint8_t* __restrict__ mask, int N, int lane) {
    RingMask M;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int i = 2 * lane + a;
#pragma unroll
        for (int wd = 0; wd < 4; ++wd) {
            uint32_t acc = 0u;
            if (i < N)
                for (int b = 0; b < 32; ++b) {
                    const int j = 32 * wd + b;
                    if (j < N && (!mask || mask[(size_t)i * N + j])) acc |= 1u << b;     // (no mask: an unmasked term of a two-term launch)
                }
            M.w[4 * a + wd] = acc;
        }
    }
    return M;
}

template <int LEVEL, bool NEAR, int RDF, int KIND, bool MASK>
__device__ __forceinline__ void ring_sweep(const RingLJ& K, const RingRdf& X, const RingMask& M, int N, int lane, const Vec3x2& q,
                                           const Vec3x2& w, Vec3x2& f, Vec3x2& g, float (&th)[MDG_MAX_THETA], Vec3x2& rq,
                                           f32x2* __restrict__ lds) {
    const int nl = (N + 1) >> 1;
    const bool vi0 = 2 * lane < N, vi1 = 2 * lane + 1 < N;
    f32x2* sqx = lds; f32x2* sqy = lds + 64; f32x2* sqz = lds + 128;
    f32x2* swx = lds + 192; f32x2* swy = lds + 256; f32x2* swz = lds + 320;
    ring_lds_fence();
    sqx[lane] = q.x; sqy[lane] = q.y; sqz[lane] = q.z;
    if constexpr (LEVEL >= 2) { swx[lane] = w.x; swy[lane] = w.y; swz[lane] = w.z; }
    ring_lds_fence();
    Vec3x2 fi = vzero(), gi = vzero(), fj = vzero(), gj = vzero(), ri = vzero(), rj = vzero();
    f32x2 S[MDG_MAX_THETA], D[MDG_MAX_THETA];                 // parameter sums: ring steps (undirected) / directed steps
#pragma unroll
    for (int k = 0; k < MDG_MAX_THETA; ++k) { S[k] = f32x2{0.f, 0.f}; D[k] = S[k]; }
    // step 0: the pair inside the lane, both directions (one copy for the histogram)
    {
        const bool v = vi0 && vi1;
        bool vm = v;
        if constexpr (MASK) vm = v && ((ring_mask_word(M.w, lane >> 4) >> (2 * (lane & 15) + 1)) & 1u);   // (i0, i1)
        ring_pair<LEVEL, NEAR, true, false, RDF, KIND>(K, X, q, w, q, w, vm, vm, v, RDF == 2 && v, fi, gi, fj, gj, D, ri, rj);
    }
    const int prev = lane < nl ? ((lane == 0 ? nl : lane) - 1) * 4 : lane * 4;     // bpermute address of lane l-1
    const int nsteps = (nl - 1) >> 1;
    int idx = lane;
    Vec3x2 qj, wj = vzero();
#pragma unroll 1
    for (int k = 1; k <= nsteps; ++k) {
        idx -= 1; idx = idx < 0 ? idx + nl : idx;
        qj.x = sqx[idx]; qj.y = sqy[idx]; qj.z = sqz[idx];
        if constexpr (LEVEL >= 1) fj = ring_move(fj, prev);
        if constexpr (LEVEL >= 2) { wj.x = swx[idx]; wj.y = swy[idx]; wj.z = swz[idx]; gj = ring_move(gj, prev); }
        if constexpr (RDF == 2) rj = ring_move(rj, prev);
        const bool vj0 = 2 * idx < N, vj1 = 2 * idx + 1 < N;
        const bool s0 = vi0 && vj0, s1 = vi1 && vj1, c0 = vi0 && vj1, c1 = vi1 && vj0;
        bool ms0 = s0, ms1 = s1, mc0 = c0, mc1 = c1;
        if constexpr (MASK) {
            const uint32_t b0 = ring_mask_word(M.w, idx >> 4) >> (2 * (idx & 15)), b1 = ring_mask_word(M.w + 4, idx >> 4) >> (2 * (idx & 15));
            ms0 = s0 && (b0 & 1u); mc0 = c0 && (b0 & 2u); mc1 = c1 && (b1 & 1u); ms1 = s1 && (b1 & 2u);
        }
        ring_pair<LEVEL, NEAR, false, true, RDF, KIND>(K, X, q, w, qj, wj, ms0, ms1, s0, s1, fi, gi, fj, gj, S, ri, rj);
        ring_pair<LEVEL, NEAR, true, true, RDF, KIND>(K, X, q, w, qj, wj, mc0, mc1, c0, c1, fi, gi, fj, gj, S, ri, rj);
    }
    if (!(nl & 1)) {
        // antipodal lanes (k = nl/2) see each other from both sides -> directed evaluation, visitors not updated
        // (the lower lane of the two feeds the histogram)
        idx -= 1; idx = idx < 0 ? idx + nl : idx;
        qj.x = sqx[idx]; qj.y = sqy[idx]; qj.z = sqz[idx];
        if constexpr (LEVEL >= 2) { wj.x = swx[idx]; wj.y = swy[idx]; wj.z = swz[idx]; }
        const bool vj0 = 2 * idx < N, vj1 = 2 * idx + 1 < N;
        const bool s0 = vi0 && vj0, s1 = vi1 && vj1, c0 = vi0 && vj1, c1 = vi1 && vj0;
        bool ms0 = s0, ms1 = s1, mc0 = c0, mc1 = c1;
        if constexpr (MASK) {
            const uint32_t b0 = ring_mask_word(M.w, idx >> 4) >> (2 * (idx & 15)), b1 = ring_mask_word(M.w + 4, idx >> 4) >> (2 * (idx & 15));
            ms0 = s0 && (b0 & 1u); mc0 = c0 && (b0 & 2u); mc1 = c1 && (b1 & 1u); ms1 = s1 && (b1 & 2u);
        }
        const bool once = RDF == 2 || 2 * lane < nl;
        Vec3x2 fu = vzero(), gu = vzero(), ru = vzero();
        ring_pair<LEVEL, NEAR, false, false, RDF, KIND>(K, X, q, w, qj, wj, ms0, ms1, s0 && once, s1 && once, fi, gi, fu, gu, D, ri, ru);
        ring_pair<LEVEL, NEAR, true, false, RDF, KIND>(K, X, q, w, qj, wj, mc0, mc1, c0 && once, c1 && once, fi, gi, fu, gu, D, ri, ru);
    }
    // the travelling accumulators are nsteps lanes ahead of their owners
    int home = lane + nsteps; home = home >= nl ? home - nl : home;
    home = (lane < nl ? home : lane) * 4;
    if constexpr (LEVEL >= 1) {
        fj = ring_move(fj, home);
        f.x = fi.x + fj.x; f.y = fi.y + fj.y; f.z = fi.z + fj.z;
    }
    if constexpr (LEVEL >= 2) {
        gj = ring_move(gj, home);
        g.x = -(gi.x + gj.x); g.y = -(gi.y + gj.y); g.z = -(gi.z + gj.z);
#pragma unroll
        for (int k = 0; k < MDG_MAX_THETA; ++k)
            th[k] = 2.f * hsum(S[k]) + hsum(D[k]);  // an undirected pair of the ring steps stands for both directions
    }
    if constexpr (RDF == 2) {
        rj = ring_move(rj, home);
        rq.x = ri.x + rj.x; rq.y = ri.y + rj.y; rq.z = ri.z + rj.z;
    }
}

// window test of the fast minimum image (see force_all_pairs): every atom within [-0.24, 1.24] cell lengths
__device__ __forceinline__ bool ring_near(const RingLJ& K, const Vec3x2& q) {
    const f32x2 sx = q.x * K.ivx, sy = q.y * K.ivy, sz = q.z * K.ivz;
    bool out = false;
    out |= !(sx.x > -0.24f && sx.x < 1.24f) | !(sx.y > -0.24f && sx.y < 1.24f);
    out |= !(sy.x > -0.24f && sy.x < 1.24f) | !(sy.y > -0.24f && sy.y < 1.24f);
    out |= !(sz.x > -0.24f && sz.x < 1.24f) | !(sz.y > -0.24f && sz.y < 1.24f);
    return __builtin_amdgcn_ballot_w64(out) == 0;
}

// RDF: compile-time mode of the kernel; with_rdf: this frame is one of the observable's frames (wave-uniform)
template <int LEVEL, int RDF, int KIND, bool MASK>
__device__ __forceinline__ void ring_force(const RingLJ& K, const RingRdf& X, const RingMask& M, bool with_rdf, int N, int lane,
                                           const Vec3x2& q, const Vec3x2& w, Vec3x2& f, Vec3x2& g, float (&th)[MDG_MAX_THETA],
                                           Vec3x2& rq, f32x2* __restrict__ lds) {
    const bool near = ring_near(K, q);
    if constexpr (RDF != 0) {
        if (with_rdf) {
            if (near) ring_sweep<LEVEL, true, RDF, KIND, MASK>(K, X, M, N, lane, q, w, f, g, th, rq, lds);
            else ring_sweep<LEVEL, false, RDF, KIND, MASK>(K, X, M, N, lane, q, w, f, g, th, rq, lds);
            return;
        }
    }
    if constexpr (LEVEL >= 1) {
        if (near) ring_sweep<LEVEL, true, 0, KIND, MASK>(K, X, M, N, lane, q, w, f, g, th, rq, lds);
        else ring_sweep<LEVEL, false, 0, KIND, MASK>(K, X, M, N, lane, q, w, f, g, th, rq, lds);
    }
}

// ---- several LJ 12-6 terms in ONE sweep (round 6): the geometry of a pair -- minimum image, d^2, 1/d^2: two thirds of a pair
// operation -- is shared by the terms; each term adds its own polynomial where its mask selects the pair and the pair is
// inside ITS cutoff, and keeps its own parameter sums.  Overlapping selections (a masked term over an unmasked one) simply add.
template <int LEVEL, bool NEAR, bool CROSS, bool JSIDE, int NT>
__device__ __forceinline__ void ring_pair_lj_multi(const RingLJ (&K)[NT], const Vec3x2& qi, const Vec3x2& wi, const Vec3x2& qj,
                                                   const Vec3x2& wj, const bool (&v0)[NT], const bool (&v1)[NT], Vec3x2& fi,
                                                   Vec3x2& gi, Vec3x2& fj, Vec3x2& gj, f32x2 (&TH)[NT][MDG_MAX_THETA]) {
    f32x2 dx = (CROSS ? qj.x.yx : qj.x) - qi.x, dy = (CROSS ? qj.y.yx : qj.y) - qi.y,
          dz = (CROSS ? qj.z.yx : qj.z) - qi.z;                                       // D = x_j - x_i
    if constexpr (NEAR) {
        dx = min_image_diag2_near(dx, K[0].ivx, K[0].hx); dy = min_image_diag2_near(dy, K[0].ivy, K[0].hy);
        dz = min_image_diag2_near(dz, K[0].ivz, K[0].hz);
    } else {
        dx = min_image_diag2(dx, K[0].ivx, K[0].hx); dy = min_image_diag2(dy, K[0].ivy, K[0].hy);
        dz = min_image_diag2(dz, K[0].ivz, K[0].hz);
    }
    const f32x2 d2 = norm2_ref2(dx, dy, dz);
    const bool nz0 = d2.x != 0.f, nz1 = d2.y != 0.f;                                  // topology.py:67
    const f32x2 r2 = {nz0 ? __builtin_amdgcn_rcpf(d2.x) : 0.f, nz1 ? __builtin_amdgcn_rcpf(d2.y) : 0.f};
    f32x2 c1 = {0.f, 0.f}, kk = {0.f, 0.f}, t6[NT], t12[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m) {
        const bool ok0 = v0[m] && (d2.x < K[m].rc2), ok1 = v1[m] && (d2.y < K[m].rc2);
        const f32x2 i2 = {ok0 ? r2.x : 0.f, ok1 ? r2.y : 0.f};                        // (0 for a rejected pair: every term below vanishes)
        const f32x2 s2 = K[m].sig2 * i2;
        const f32x2 s6 = s2 * s2 * s2;
        const f32x2 s12 = s6 * s6;
        c1 += (K[m].m1a * s6 - K[m].m1b * s12) * i2;
        if constexpr (LEVEL >= 2) {
            kk += (K[m].kb * s12 - K[m].ka * s6) * (i2 * i2);
            t6[m] = s6 * i2; t12[m] = s12 * i2;
        }
    }
    fi.x += c1 * dx; fi.y += c1 * dy; fi.z += c1 * dz;                                // F_i += (phi'/r) D
    if constexpr (JSIDE) {
        if constexpr (CROSS) {
            fj.x = __builtin_elementwise_fma(-c1.yx, dx.yx, fj.x); fj.y = __builtin_elementwise_fma(-c1.yx, dy.yx, fj.y);
            fj.z = __builtin_elementwise_fma(-c1.yx, dz.yx, fj.z);
        } else {
            fj.x -= c1 * dx; fj.y -= c1 * dy; fj.z -= c1 * dz;
        }
    }
    if constexpr (LEVEL >= 2) {
        const f32x2 ax = wi.x - (CROSS ? wj.x.yx : wj.x), ay = wi.y - (CROSS ? wj.y.yx : wj.y),
                    az = wi.z - (CROSS ? wj.z.yx : wj.z);
        const f32x2 b = dx * ax + dy * ay + dz * az;                                  // w_ij . D
        const f32x2 k2 = kk * b;
        const f32x2 tx = __builtin_elementwise_fma(k2, dx, c1 * ax), ty = __builtin_elementwise_fma(k2, dy, c1 * ay),
                    tz = __builtin_elementwise_fma(k2, dz, c1 * az);                  // -(H w) contribution
        gi.x += tx; gi.y += ty; gi.z += tz;
        if constexpr (JSIDE) {
            gj.x -= CROSS ? tx.yx : tx; gj.y -= CROSS ? ty.yx : ty; gj.z -= CROSS ? tz.yx : tz;
        }
#pragma unroll
        for (int m = 0; m < NT; ++m) {                    // (a directed pair counts half of an undirected one: ONE set of sums, th = 2 S)
            const f32x2 bw = JSIDE ? b : 0.5f * b;
            TH[m][0] += t6[m] * bw; TH[m][1] += t12[m] * bw;
        }
    }
}

// ring_sweep for NT LJ 12-6 terms at once (no fused observable): the ring of ring_sweep, per-term mask bits and sums
template <int LEVEL, bool NEAR, int NT>
__device__ __forceinline__ void ring_sweep_lj_multi(const RingLJ (&K)[NT], const RingMask (&M)[NT], int N, int lane, const Vec3x2& q,
                                                    const Vec3x2& w, Vec3x2& f, Vec3x2& g, float (&th)[NT][MDG_MAX_THETA],
                                                    f32x2* __restrict__ lds) {
    const int nl = (N + 1) >> 1;
    const bool vi0 = 2 * lane < N, vi1 = 2 * lane + 1 < N;
    f32x2* sqx = lds; f32x2* sqy = lds + 64; f32x2* sqz = lds + 128;
    f32x2* swx = lds + 192; f32x2* swy = lds + 256; f32x2* swz = lds + 320;
    ring_lds_fence();
    sqx[lane] = q.x; sqy[lane] = q.y; sqz[lane] = q.z;
    if constexpr (LEVEL >= 2) { swx[lane] = w.x; swy[lane] = w.y; swz[lane] = w.z; }
    ring_lds_fence();
    Vec3x2 fi = vzero(), gi = vzero(), fj = vzero(), gj = vzero();
    f32x2 S[NT][MDG_MAX_THETA];
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int k = 0; k < MDG_MAX_THETA; ++k) S[m][k] = f32x2{0.f, 0.f};
    {   // step 0: the pair inside the lane, both directions
        const bool v = vi0 && vi1;
        bool vm[NT];
#pragma unroll
        for (int m = 0; m < NT; ++m) vm[m] = v && ((ring_mask_word(M[m].w, lane >> 4) >> (2 * (lane & 15) + 1)) & 1u);
        ring_pair_lj_multi<LEVEL, NEAR, true, false, NT>(K, q, w, q, w, vm, vm, fi, gi, fj, gj, S);
    }
    const int prev = lane < nl ? ((lane == 0 ? nl : lane) - 1) * 4 : lane * 4;
    const int nsteps = (nl - 1) >> 1;
    int idx = lane;
    Vec3x2 qj, wj = vzero();
#pragma unroll 1
    for (int k = 1; k <= nsteps; ++k) {
        idx -= 1; idx = idx < 0 ? idx + nl : idx;
        qj.x = sqx[idx]; qj.y = sqy[idx]; qj.z = sqz[idx];
        fj = ring_move(fj, prev);
        if constexpr (LEVEL >= 2) { wj.x = swx[idx]; wj.y = swy[idx]; wj.z = swz[idx]; gj = ring_move(gj, prev); }
        const bool vj0 = 2 * idx < N, vj1 = 2 * idx + 1 < N;
        const bool s0 = vi0 && vj0, s1 = vi1 && vj1, c0 = vi0 && vj1, c1 = vi1 && vj0;
        bool ms0[NT], ms1[NT], mc0[NT], mc1[NT];
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            const uint32_t b0 = ring_mask_word(M[m].w, idx >> 4) >> (2 * (idx & 15)), b1 = ring_mask_word(M[m].w + 4, idx >> 4) >> (2 * (idx & 15));
            ms0[m] = s0 && (b0 & 1u); mc0[m] = c0 && (b0 & 2u); mc1[m] = c1 && (b1 & 1u); ms1[m] = s1 && (b1 & 2u);
        }
        ring_pair_lj_multi<LEVEL, NEAR, false, true, NT>(K, q, w, qj, wj, ms0, ms1, fi, gi, fj, gj, S);
        ring_pair_lj_multi<LEVEL, NEAR, true, true, NT>(K, q, w, qj, wj, mc0, mc1, fi, gi, fj, gj, S);
    }
    if (!(nl & 1)) {                                                  // antipodal lanes: directed evaluation, visitors not updated
        idx -= 1; idx = idx < 0 ? idx + nl : idx;
        qj.x = sqx[idx]; qj.y = sqy[idx]; qj.z = sqz[idx];
        if constexpr (LEVEL >= 2) { wj.x = swx[idx]; wj.y = swy[idx]; wj.z = swz[idx]; }
        const bool vj0 = 2 * idx < N, vj1 = 2 * idx + 1 < N;
        const bool s0 = vi0 && vj0, s1 = vi1 && vj1, c0 = vi0 && vj1, c1 = vi1 && vj0;
        bool ms0[NT], ms1[NT], mc0[NT], mc1[NT];
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            const uint32_t b0 = ring_mask_word(M[m].w, idx >> 4) >> (2 * (idx & 15)), b1 = ring_mask_word(M[m].w + 4, idx >> 4) >> (2 * (idx & 15));
            ms0[m] = s0 && (b0 & 1u); mc0[m] = c0 && (b0 & 2u); mc1[m] = c1 && (b1 & 1u); ms1[m] = s1 && (b1 & 2u);
        }
        Vec3x2 fu = vzero(), gu = vzero();
        ring_pair_lj_multi<LEVEL, NEAR, false, false, NT>(K, q, w, qj, wj, ms0, ms1, fi, gi, fu, gu, S);
        ring_pair_lj_multi<LEVEL, NEAR, true, false, NT>(K, q, w, qj, wj, mc0, mc1, fi, gi, fu, gu, S);
    }
    int home = lane + nsteps; home = home >= nl ? home - nl : home;
    home = (lane < nl ? home : lane) * 4;
    fj = ring_move(fj, home);
    f.x = fi.x + fj.x; f.y = fi.y + fj.y; f.z = fi.z + fj.z;
    if constexpr (LEVEL >= 2) {
        gj = ring_move(gj, home);
        g.x = -(gi.x + gj.x); g.y = -(gi.y + gj.y); g.z = -(gi.z + gj.z);
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
            for (int k = 0; k < MDG_MAX_THETA; ++k) th[m][k] = 2.f * hsum(S[m][k]);
    }
}

// NT terms of one pair form (round 6: the species mixtures of scripts/fit_mix.py -- index_tuple stacks, torchmd/interface.py:
// 228-260): LJ 12-6 terms share ONE sweep (ring_sweep_lj_multi); other LJ-family powers take one ring sweep per term with the
// term's constants and its 128-bit mask rows, forces and Hessian.w added up, the parameter sums kept per term.  The fused
// observable rides on the first term's sweep only (its flags are unmasked).
template <int LEVEL, int RDF, int KIND, bool MASK, int NT>
__device__ __forceinline__ void ring_force_terms(const RingLJ (&K)[NT], const RingRdf& X, const RingMask (&M)[NT], bool with_rdf, int N,
                                                 int lane, const Vec3x2& q, const Vec3x2& w, Vec3x2& f, Vec3x2& g,
                                                 float (&th)[NT][MDG_MAX_THETA], Vec3x2& rq, f32x2* __restrict__ lds) {
    if constexpr (NT > 1 && KIND == KIND_LJ126 && RDF == 0) {
        // LJ 12-6 terms: ONE sweep, the pair geometry shared (ring_sweep_lj_multi)
        if constexpr (LEVEL >= 1) {
            if (ring_near(K[0], q)) ring_sweep_lj_multi<LEVEL, true, NT>(K, M, N, lane, q, w, f, g, th, lds);
            else ring_sweep_lj_multi<LEVEL, false, NT>(K, M, N, lane, q, w, f, g, th, lds);
        }
        return;
    }
    ring_force<LEVEL, RDF, KIND, MASK>(K[0], X, M[0], with_rdf, N, lane, q, w, f, g, th[0], rq, lds);
    if constexpr (NT > 1 && LEVEL >= 1) {
#pragma unroll
        for (int m = 1; m < NT; ++m) {
            Vec3x2 f2 = vzero(), g2 = vzero(), r2 = vzero();
            ring_force<LEVEL, 0, KIND, MASK>(K[m], X, M[m], false, N, lane, q, w, f2, g2, th[m], r2, lds);
            f.x += f2.x; f.y += f2.y; f.z += f2.z;
            if constexpr (LEVEL >= 2) { g.x += g2.x; g.y += g2.y; g.z += g2.z; }
        }
    }
}

// ------------------------------------------------------------------------------------------------ frame I/O
// AoS [N,3] frames in HBM: lane l reads / writes the six consecutive floats of its two atoms.
__device__ __forceinline__ Vec3x2 ring_load(const float* __restrict__ src, int N, int lane) {
    Vec3x2 r = vzero();
    const float* p = src + 6 * lane;
    if (2 * lane + 1 < N) {
        const f32x2 a = *reinterpret_cast<const f32x2*>(p), b = *reinterpret_cast<const f32x2*>(p + 2),
                    c = *reinterpret_cast<const f32x2*>(p + 4);
        r.x = f32x2{a.x, b.y}; r.y = f32x2{a.y, c.x}; r.z = f32x2{b.x, c.y};
    } else if (2 * lane < N) {
        r.x.x = p[0]; r.y.x = p[1]; r.z.x = p[2];
    }
    return r;
}
__device__ __forceinline__ void ring_store(float* __restrict__ dst, const Vec3x2& v, int N, int lane) {
    float* p = dst + 6 * lane;
    if (2 * lane + 1 < N) {
        *reinterpret_cast<f32x2*>(p) = f32x2{v.x.x, v.y.x};
        *reinterpret_cast<f32x2*>(p + 2) = f32x2{v.z.x, v.x.y};
        *reinterpret_cast<f32x2*>(p + 4) = f32x2{v.y.y, v.z.y};
    } else if (2 * lane < N) {
        p[0] = v.x.x; p[1] = v.y.x; p[2] = v.z.x;
    }
}

// ------------------------------------------------------------------------------------------------ bath
// Nose-Hoover chain, entry k in lane k (md.py:234-236): same formulas as bath_rhs / bath_vjp
__device__ __forceinline__ float ring_bath_rhs(const TrajArgs& A, int lane, float Qk, float pv, float ke) {
    const int C = A.prm.n_chains;
    const float T = A.prm.T;
    const float dn = row_down(pv), dnQ = row_down(Qk), up = row_up(pv), upQ = row_up(Qk);
    const float ta = lane == 0 ? 2.f * (ke - T * A.prm.n_dof * 0.5f) : dn * dn / dnQ - T;
    const float tb = lane == C - 1 ? 0.f : up * pv / upQ;
    return lane < C ? ta - tb : 0.f;
}
__device__ __forceinline__ float ring_bath_vjp(const TrajArgs& A, int lane, float Qk, float pv, float lp, float slv) {
    const int C = A.prm.n_chains;
    // (every cross-lane read happens here, with all lanes active: a DPP move inside a divergent branch would read
    //  zeros from the lanes that took the other side)
    const float dnlp = row_down(lp), dnpv = row_down(pv), uppv = row_up(pv), upQ = row_up(Qk), uplp = row_up(lp);
    const float t1 = lane == 0 ? -slv / Qk : -dnlp * dnpv / Qk;
    const float t2 = lane == C - 1 ? 0.f : -lp * uppv / upQ + 2.f * pv * uplp / Qk;
    return lane < C ? t1 + t2 : 0.f;
}

// Q[lane] (a chain of selects: indexing the by-value kernel argument with a run-time index would put it in scratch)
__device__ __forceinline__ float ring_chain_mass(const TrajArgs& A, int lane) {
    float Qk = 1.f;
#pragma unroll
    for (int c = 0; c < MDG_MAX_CHAINS; ++c)
        if (lane == c && c < A.prm.n_chains) Qk = A.prm.Q[c];
    return Qk;
}

__device__ __forceinline__ float ring_dot(const Vec3x2& a, const Vec3x2& b) {      // per-lane partial of sum a.b
    return hsum(a.x * b.x + a.y * b.y + a.z * b.z);
}

// ------------------------------------------------------------------------------------------------ fused RDF
// Device-side description of the fused observable (host: MdgRdfFuse + the fine-grid plan of csrc/rdf.hip)
struct RingRdfArgs {
    const float* mu; int nbins;             // equally spaced centres (device)
    float rc;                               // the observable's pair cutoff
    int f_start, f_stride;                  // frames f_start, f_start + f_stride, ... of every replica
    float reach, inv_h; int nfine;          // forward: fine integer histogram
    uint32_t* ghist;                        //          [nfine] global, zeroed by the host
    const float4* tab; int reach_bins;      // adjoint: cell cubics of dL/dd (rdf_bwd_table_kernel), R of the grid
};
__device__ __forceinline__ bool ring_frame_selected(const RingRdfArgs& F, int k) {
    return k >= F.f_start && (k - F.f_start) % F.f_stride == 0;
}

// ------------------------------------------------------------------------------------------------ forward
// RDF = false: one wave (= one replica) per workgroup.  RDF = true: sixteen waves share the workgroup's fine
// histogram in LDS and stride over the replicas (persistent grid: the histogram is merged into HBM once per
// workgroup); the waves are otherwise independent.
template <bool RDF, int KIND, bool MASK = false, int NT = 1>
__global__ __launch_bounds__(RDF ? 1024 : 64) void traj_fwd_ring_kernel(const TrajArgs A, const RingRdfArgs F) {
    static_assert(NT == 1 || (MASK && KIND != KIND_TABLE), "several terms: masked built-in forms");
    extern __shared__ __attribute__((aligned(16))) float smr[];
    const int N = A.prm.n_atoms, T = A.prm.n_frames, C = A.prm.n_chains;
    const bool nhc = A.prm.ensemble == 0;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6, N3 = 3 * N;
    RingLJ K[NT];
    RingMask M[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m) {
        K[m] = ring_constants(A, m);
        M[m] = RingMask{};
        if constexpr (MASK) M[m] = ring_mask_load(A.terms.t[m].mask, N, lane);
    }
    const int nf2 = RDF ? (F.nfine + 1) & ~1 : 0;
    uint32_t* hist = reinterpret_cast<uint32_t*>(smr);
    f32x2* lds = reinterpret_cast<f32x2*>(smr + nf2) + wid * 3 * 64;
    RingRdf X{};
    if constexpr (KIND == KIND_TABLE) {
        // (one wave per workgroup: the unfused launch) the nodes of the tabulated pair model, behind the wave's ring buffers
        const MdgPairTerm& t0 = A.terms.t[0];
        float2* ttab = reinterpret_cast<float2*>(smr + nf2 + nw * 3 * 64 * 2);
        int32_t* tflag = reinterpret_cast<int32_t*>(ttab + t0.p);
        const float* thp = A.theta + t0.theta_off;
        for (int g = threadIdx.x; g < t0.p; g += blockDim.x) ttab[g] = make_float2(thp[2 * g], thp[2 * g + 1]);
        if (threadIdx.x == 0) *tflag = 0;
        X.ttab = ttab; X.tflag = tflag; X.tu0 = t0.a; X.tinv_du = 1.f / t0.phi; X.ttmax = (float)(t0.p - 1) - 1e-3f;
        __syncthreads();
    }
    if constexpr (RDF) {
        for (int m = threadIdx.x; m < F.nfine; m += blockDim.x) hist[m] = 0u;
        __syncthreads();
        const float lo = F.mu[0] - F.reach;                      // lower edge of the fine grid
        X.hist = hist; X.inv_h = F.inv_h; X.tlo = -lo * F.inv_h;
        X.fmax = fminf((float)F.nfine, fmaf(F.rc, F.inv_h, X.tlo));
    }
    f32x2 ms = {1.f, 1.f};                                       // (absent atoms: unit mass, zero state)
    if (2 * lane < N) ms.x = A.mass[2 * lane];
    if (2 * lane + 1 < N) ms.y = A.mass[2 * lane + 1];
    const f32x2 ims = rcp2(ms);
    const float Qk = ring_chain_mass(A, lane);
    const float iQ0 = 1.f / A.prm.Q[0];
    for (int rep = blockIdx.x * nw + wid; rep < A.prm.n_rep; rep += gridDim.x * nw) {
        const size_t fr = (size_t)rep * T;
        Vec3x2 q = ring_load(A.q0 + (size_t)rep * N3, N, lane), v = ring_load(A.v0 + (size_t)rep * N3, N, lane);
        float pv = 0.f;
        if (nhc && lane < C) pv = A.pv0[(size_t)rep * C + lane];
        // frame 0 = inputs (tinydiffeq.py:63)
        ring_store(A.q_t + fr * N3, q, N, lane);
        ring_store(A.v_t + fr * N3, v, N, lane);
        if (nhc && lane < C) A.pv_t[fr * C + lane] = pv;
        Vec3x2 f, gu, ru, wu = vzero();
        float tu[NT][MDG_MAX_THETA];
        // (every force evaluation of the forward pass is at the positions of a stored frame: the RDF rides along)
        ring_force_terms<1, RDF ? 1 : 0, KIND, MASK, NT>(K, X, M, RDF && ring_frame_selected(F, 0), N, lane, q, wu, f, gu, tu, ru, lds);
        for (int k = 0; k + 1 < T; ++k) {
            const float dt = A.t[k + 1] - A.t[k];
            // ---- first RHS at y_k (force cached), half kick + drift       sovlers.py:111-118
            float pb = 0.f, pv0 = 0.f;
            Vec3x2 vh;
            if (nhc) {
                const Vec3x2 p{v.x * ms, v.y * ms, v.z * ms};
                const float ke = 0.5f * wave_sum(ring_dot(p, v));
                pb = ring_bath_rhs(A, lane, Qk, pv, ke);
                pv0 = lane0(pv);
                const float c0 = pv0 * iQ0;
                vh.x = 0.5f * ((f.x - c0 * p.x) * ims) * dt;
                vh.y = 0.5f * ((f.y - c0 * p.y) * ims) * dt;
                vh.z = 0.5f * ((f.z - c0 * p.z) * ims) * dt;
            } else {
                vh.x = 0.5f * f.x * dt; vh.y = 0.5f * f.y * dt; vh.z = 0.5f * f.z * dt;      // md.py:145-148 (no 1/m)
            }
            q.x = q.x + (v.x + vh.x) * dt; q.y = q.y + (v.y + vh.y) * dt; q.z = q.z + (v.z + vh.z) * dt;
            const float ph = 0.5f * pb * dt, pvh = pv + ph;
            // ---- second RHS at (v + vh, q1, pv + ph)                      sovlers.py:121-125
            ring_force_terms<1, RDF ? 1 : 0, KIND, MASK, NT>(K, X, M, RDF && ring_frame_selected(F, k + 1), N, lane, q, wu, f, gu, tu, ru, lds);
            const Vec3x2 vv{v.x + vh.x, v.y + vh.y, v.z + vh.z};
            if (nhc) {
                const Vec3x2 p{vv.x * ms, vv.y * ms, vv.z * ms};
                const float ke = 0.5f * wave_sum(ring_dot(p, vv));
                const float b1 = ring_bath_rhs(A, lane, Qk, pvh, ke);
                const float c0 = lane0(pvh) * iQ0;
                pv = pv + (ph + 0.5f * b1 * dt);
                v.x = v.x + (vh.x + 0.5f * ((f.x - c0 * p.x) * ims) * dt);
                v.y = v.y + (vh.y + 0.5f * ((f.y - c0 * p.y) * ims) * dt);
                v.z = v.z + (vh.z + 0.5f * ((f.z - c0 * p.z) * ims) * dt);
            } else {
                v.x = v.x + (vh.x + 0.5f * f.x * dt); v.y = v.y + (vh.y + 0.5f * f.y * dt); v.z = v.z + (vh.z + 0.5f * f.z * dt);
            }
            ring_store(A.q_t + (fr + k + 1) * N3, q, N, lane);
            ring_store(A.v_t + (fr + k + 1) * N3, v, N, lane);
            if (nhc && lane < C) A.pv_t[(fr + k + 1) * C + lane] = pv;
        }
        if (A.nonfinite) {
            const bool bad = !(isfinite(hsum(q.x + q.y + q.z)) && isfinite(hsum(v.x + v.y + v.z)));
            int below = 0;
            if constexpr (KIND == KIND_TABLE) {                           // (bit 1: a live pair below the table's first node)
                ring_lds_fence();
                below = *X.tflag;
                ring_lds_fence();
                if (lane == 0) *X.tflag = 0;
            }
            const int nf = __builtin_amdgcn_ballot_w64(bad) != 0 ? 1 : 0;
            if ((nf | below) && lane == 0) A.nonfinite[rep] = nf | below;
        }
    }
    if constexpr (RDF) {
        __syncthreads();
        for (int m = threadIdx.x; m < F.nfine; m += blockDim.x) {
            const uint32_t c = hist[m];
            if (c) atomicAdd(&F.ghist[m], c);
        }
    }
}

// parameter gradient of one augmented evaluation: wave totals of the sweep's sums, weighted with the interval
// (NVE: both half steps carry the first evaluation's term, sovlers.py:82,101)
template <int KIND>
__device__ __forceinline__ void ring_theta(const RingLJ& K, const float (&th)[MDG_MAX_THETA], float (&gth)[MDG_MAX_THETA],
                                           float h, bool nve) {
    if constexpr (KIND == KIND_TABLE) {
        // (the table gradient is scattered by the sweep itself)
    } else if constexpr (KIND == KIND_LJ126) {
        const float t6 = wave_sum(th[0]), t12 = wave_sum(th[1]);
        const float gs = K.tsa * t6 - K.tsb * t12, ge = K.tea * t6 - K.teb * t12;
        gth[0] += nve ? (gs * 0.5f * h) * 2.f : gs * h;
        gth[1] += nve ? (ge * 0.5f * h) * 2.f : ge * h;
    } else {
#pragma unroll
        for (int k = 0; k < kind_ntheta(KIND); ++k) {
            const float tk = wave_sum(th[k]);
            gth[k] += nve ? (tk * 0.5f * h) * 2.f : tk * h;
        }
    }
}

// ------------------------------------------------------------------------------------------------ adjoint
// RDF = true: the frame gradients of the fused observable are produced here -- in the first augmented evaluation
// of interval i (which sits at frame i) for frames T-1 .. 1, and in one geometry-only sweep for frame 0 -- and
// added to lam_q where the adjoint adds the incoming g_q (sovlers.py:249, :286).
// KIND_TABLE: RING_TABLE_WAVES replicas (one wave each) per workgroup share the nodes AND one pair of gradient planes --
// per-wave planes (24 KB at 1 024 nodes) would leave five waves per CU where the registers allow eight.  The workgroup's
// table gradient goes to the adj_theta row of its first replica, the rows of its other replicas are zero: only the sum over
// replicas of a tabulated kind's rows is defined (what the caller forms, ops.FusedTrajFn.backward).
constexpr int RING_TABLE_WAVES = 8;

template <bool RDF, int KIND, bool MASK = false, int NT = 1>
__global__ __launch_bounds__(KIND == KIND_TABLE ? 64 * RING_TABLE_WAVES : 64) void traj_adj_ring_kernel(const TrajArgs A, const RingRdfArgs F) {
    static_assert(NT == 1 || (MASK && KIND != KIND_TABLE), "several terms: masked built-in forms");
    extern __shared__ __attribute__((aligned(16))) float smr[];
    const int N = A.prm.n_atoms, T = A.prm.n_frames, C = A.prm.n_chains;
    const bool nhc = A.prm.ensemble == 0;
    constexpr int NWV = KIND == KIND_TABLE ? RING_TABLE_WAVES : 1;
    const int wid = KIND == KIND_TABLE ? (int)(threadIdx.x >> 6) : 0, lane = threadIdx.x & 63, N3 = 3 * N;
    // (a wave beyond the last replica of the last workgroup repeats the last replica without accumulating or storing)
    // (the other kinds launch one wave per replica: `live` is a compile-time true there, the code of round 4)
    const bool live = KIND != KIND_TABLE || (int)(blockIdx.x * NWV + wid) < A.prm.n_rep;
    const int rep = KIND != KIND_TABLE ? (int)blockIdx.x : (live ? (int)(blockIdx.x * NWV + wid) : A.prm.n_rep - 1);
    RingLJ K[NT];
    RingMask M[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m) {
        K[m] = ring_constants(A, m);
        M[m] = RingMask{};
        if constexpr (MASK) M[m] = ring_mask_load(A.terms.t[m].mask, N, lane);
    }
    RingRdf X{};
    int ncell = 0;
    if constexpr (RDF) {
        // the cells of rdf_bwd_fine_kernel's table (8 per centre spacing over the same distances), equally spaced in d^2
        float hu;
        rdf_u_grid(F.mu, F.nbins, F.reach_bins, X.ulo, hu, ncell);
        float4* tab = reinterpret_cast<float4*>(smr);
        for (int n = lane; n < ncell; n += 64) tab[n] = F.tab[n];
        X.tab = tab; X.inv_hu = 1.0f / hu;
        X.tmax = fminf((float)ncell, (F.rc * F.rc - X.ulo) * X.inv_hu);
        __syncthreads();
    }
    f32x2* lds = reinterpret_cast<f32x2*>(smr + 4 * ncell) + wid * 6 * 64;
    if constexpr (KIND == KIND_TABLE) {
        // the nodes and the workgroup's two gradient planes behind the waves' ring buffers (6 x 64 f32x2 each)
        const MdgPairTerm& t0 = A.terms.t[0];
        float2* ttab = reinterpret_cast<float2*>(smr + 4 * ncell + NWV * 6 * 64 * 2);
        unsigned long long* tg = reinterpret_cast<unsigned long long*>(ttab + t0.p);        // [2 p] int64 words
        int32_t* tflag = reinterpret_cast<int32_t*>(tg + 2 * t0.p);
        const float* thp = A.theta + t0.theta_off;
        for (int g = threadIdx.x; g < t0.p; g += blockDim.x) ttab[g] = make_float2(thp[2 * g], thp[2 * g + 1]);
        for (int g = threadIdx.x; g < 2 * t0.p; g += blockDim.x) tg[g] = 0ull;
        if (threadIdx.x == 0) *tflag = 0;
        X.ttab = ttab; X.tg64 = tg; X.tflag = tflag; X.tgw = 0.f;
        // one word can receive a contribution from every pair evaluation of the workgroup's replicas: one accumulating
        // evaluation per interval, N (N - 1) / 2 pairs, two ends each
        X.tlim = fx64_limit((double)NWV * (double)(T > 1 ? T - 1 : 1) * (double)N * (double)N);
        X.tu0 = t0.a; X.tinv_du = 1.f / t0.phi; X.ttmax = (float)(t0.p - 1) - 1e-3f;
        __syncthreads();
    }
    const size_t fr = (size_t)rep * T;
    f32x2 ms = {1.f, 1.f};
    if (2 * lane < N) ms.x = A.mass[2 * lane];
    if (2 * lane + 1 < N) ms.y = A.mass[2 * lane + 1];
    const f32x2 ims = rcp2(ms);
    const float Qk = ring_chain_mass(A, lane);
    const float iQ0 = 1.f / A.prm.Q[0];
    // lam = dL/dy_{T-1}                                            sovlers.py:249
    Vec3x2 lv = A.g_v ? ring_load(A.g_v + (fr + T - 1) * N3, N, lane) : vzero();
    Vec3x2 lq = A.g_q ? ring_load(A.g_q + (fr + T - 1) * N3, N, lane) : vzero();
    float lp = (nhc && lane < C && A.g_pv) ? A.g_pv[(fr + T - 1) * C + lane] : 0.f;
    float gth[NT][MDG_MAX_THETA];
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int k = 0; k < MDG_MAX_THETA; ++k) gth[m][k] = 0.f;
    for (int i = T - 1; i >= 1; --i) {
        const float h = A.t[i] - A.t[i - 1];
        Vec3x2 q = ring_load(A.q_t + (fr + i) * N3, N, lane), v = ring_load(A.v_t + (fr + i) * N3, N, lane);
        float pv = (nhc && lane < C) ? A.pv_t[(fr + i) * C + lane] : 0.f;
        Vec3x2 w, f, dq, rq = vzero(), ru;
        float th[NT][MDG_MAX_THETA];
        // ---------------- first augmented evaluation at (y_i, lam)
        if (nhc) { w.x = lv.x * ims; w.y = lv.y * ims; w.z = lv.z * ims; } else w = lv;
        const bool with_rdf = RDF && ring_frame_selected(F, i);
        // (table kind: the parameter term of an interval comes from the midpoint evaluation for NHC, sovlers.py:160, and
        //  from this first one for NVE, :82,101 -- both with total weight h)
        if constexpr (KIND == KIND_TABLE) X.tgw = (nhc || !live) ? 0.f : 0.5f * h * A.terms.t[0].c;
        ring_force_terms<2, RDF ? 2 : 0, KIND, MASK, NT>(K, X, M, with_rdf, N, lane, q, w, f, dq, th, rq, lds);
        if (with_rdf) { lq.x += rq.x; lq.y += rq.y; lq.z += rq.z; }       // dL/dq_t[i] of the fused observable
        Vec3x2 lvh, lqh;
        if (nhc) {
            const Vec3x2 p{v.x * ms, v.y * ms, v.z * ms};
            const float ke = 0.5f * wave_sum(ring_dot(p, v)), slv = wave_sum(ring_dot(lv, v));
            const float pv0 = lane0(pv), lp0 = lane0(lp), c0 = pv0 * iQ0;
            const float pb = ring_bath_rhs(A, lane, Qk, pv, ke), gp = ring_bath_vjp(A, lane, Qk, pv, lp, slv);
            const float hh = 0.5f * h;
#define MDG_RING_HALF(c)                                                                     \
            {                                                                                \
                const f32x2 a = (f.c - c0 * p.c) * ims;                                      \
                const f32x2 Gv = -c0 * lv.c + lq.c + 2.f * ms * v.c * lp0;                   \
                const f32x2 vhalf = 0.5f * (-a) * h;                  /* sovlers.py:132 */   \
                q.c = q.c + (v.c + vhalf) * h;                        /* :138 (quirk)   */   \
                v.c = v.c + vhalf;                                                           \
                lvh.c = lv.c + Gv * hh;                               /* :141 */             \
                lqh.c = lq.c + dq.c * hh;                             /* :142 */             \
            }
            MDG_RING_HALF(x) MDG_RING_HALF(y) MDG_RING_HALF(z)
#undef MDG_RING_HALF
            const float lph = lp + gp * hh;                           // :143
            pv = pv + 0.5f * (-pb) * h;                               // :135
            // ---------------- midpoint evaluation                    :147-150
            w.x = lvh.x * ims; w.y = lvh.y * ims; w.z = lvh.z * ims;
            if constexpr (KIND == KIND_TABLE) X.tgw = live ? 0.5f * h * A.terms.t[0].c : 0.f;
            ring_force_terms<2, 0, KIND, MASK, NT>(K, X, M, false, N, lane, q, w, f, dq, th, ru, lds);
            const float slm = wave_sum(ring_dot(lvh, v));
            const float cm = lane0(pv) * iQ0, lpm0 = lane0(lph);
            const float gpm = ring_bath_vjp(A, lane, Qk, pv, lph, slm);
            const Vec3x2 gv = A.g_v ? ring_load(A.g_v + (fr + i - 1) * N3, N, lane) : vzero();
            const Vec3x2 gq = A.g_q ? ring_load(A.g_q + (fr + i - 1) * N3, N, lane) : vzero();
#define MDG_RING_FULL(c)                                                                     \
            {                                                                                \
                const f32x2 Gv = -cm * lvh.c + lqh.c + 2.f * ms * v.c * lpm0;                \
                lv.c = (lv.c + Gv * h) + gv.c;                        /* :156, :286 */       \
                lq.c = (lq.c + dq.c * h) + gq.c;                      /* :157 */             \
            }
            MDG_RING_FULL(x) MDG_RING_FULL(y) MDG_RING_FULL(z)
#undef MDG_RING_FULL
            float nlp = lp + gpm * h;                                 // :158
            if (lane < C && A.g_pv) nlp += A.g_pv[(fr + i - 1) * C + lane];
            lp = nlp;
#pragma unroll
            for (int m = 0; m < NT; ++m) ring_theta<KIND>(K[m], th[m], gth[m], h, false);     // :160
        } else {
            // verlet_update backward branch                          sovlers.py:42-101
#pragma unroll
            for (int m = 0; m < NT; ++m) ring_theta<KIND>(K[m], th[m], gth[m], h, true);      // :82,101
#define MDG_RING_NVE(c)                                                                      \
            {                                                                                \
                const f32x2 vhalf = v.c - 0.5f * (-f.c) * h;          /* :49-50 */           \
                q.c = q.c - vhalf * h;                                /* :51-52 */           \
                v.c = vhalf;                                                                 \
                const f32x2 dx = dq.c * h * 0.5f;                     /* :71 */              \
                lvh.c = lv.c + (lq.c + dx) * h;                       /* :72 */              \
                lqh.c = lq.c + dx;                                                           \
            }
            MDG_RING_NVE(x) MDG_RING_NVE(y) MDG_RING_NVE(z)
#undef MDG_RING_NVE
            if constexpr (KIND == KIND_TABLE) X.tgw = 0.f;
            ring_force_terms<2, 0, KIND, MASK, NT>(K, X, M, false, N, lane, q, lvh, f, dq, th, ru, lds);
            const Vec3x2 gv = A.g_v ? ring_load(A.g_v + (fr + i - 1) * N3, N, lane) : vzero();
            const Vec3x2 gq = A.g_q ? ring_load(A.g_q + (fr + i - 1) * N3, N, lane) : vzero();
            lv.x = lvh.x + gv.x; lv.y = lvh.y + gv.y; lv.z = lvh.z + gv.z;
            lq.x = (lqh.x + dq.x * h * 0.5f) + gq.x;                  // :100
            lq.y = (lqh.y + dq.y * h * 0.5f) + gq.y;
            lq.z = (lqh.z + dq.z * h * 0.5f) + gq.z;
        }
    }
    if constexpr (RDF) {
        if (ring_frame_selected(F, 0)) {                              // frame 0: no force evaluation there
            const Vec3x2 q = ring_load(A.q_t + fr * N3, N, lane), wu = vzero();
            Vec3x2 fu, gu, rq = vzero();
            float tu[MDG_MAX_THETA];
            ring_force<0, 2, KIND, MASK>(K[0], X, M[0], true, N, lane, q, wu, fu, gu, tu, rq, lds);
            lq.x += rq.x; lq.y += rq.y; lq.z += rq.z;
        }
    }
    if (live) {
        ring_store(A.adj_v0 + (size_t)rep * N3, lv, N, lane);
        ring_store(A.adj_q0 + (size_t)rep * N3, lq, N, lane);
        if (nhc && lane < C && A.adj_pv0) A.adj_pv0[(size_t)rep * C + lane] = lp;
    }
    if constexpr (KIND == KIND_TABLE) {
        // table gradient of the workgroup's replicas: fixed point -> float into the row of its first replica (zeros into
        // the others'); an out-of-range contribution poisons the output (the host re-scales and reports it), as in
        // traj_adj_kernel
        __syncthreads();
        if (A.adj_theta) {
            const int M2 = 2 * A.terms.t[0].p, KT = A.terms.n_theta_total;
            const bool worst = (*X.tflag & 4) != 0;
            const double inv = 1.0 / (double)A.terms.t[0].c;
            float* out = A.adj_theta + (size_t)(blockIdx.x * NWV) * KT + A.terms.t[0].theta_off;
            for (int g = threadIdx.x; g < M2; g += blockDim.x)
                out[g] = worst ? __builtin_inff() : (float)((double)(long long)X.tg64[g] * inv);
            if (wid > 0 && live) {
                float* mine = A.adj_theta + (size_t)rep * KT + A.terms.t[0].theta_off;
                for (int g = lane; g < M2; g += 64) mine[g] = 0.f;
            }
        }
        return;
    }
    if (lane == 0 && A.adj_theta) {
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            float* out = A.adj_theta + (size_t)rep * A.terms.n_theta_total + A.terms.t[m].theta_off;
#pragma unroll
            for (int k = 0; k < MDG_MAX_THETA; ++k)
                if (k < A.terms.t[m].n_theta) out[k] = gth[m][k];
        }
    }
}
