// Fused SchNet interaction block for the MD path (K9 + K10 in one kernel family):
//
//   W_e = Dense2( ssp( Dense1( smear(d_e) ) ) )                 nff/nn/modules.py:531-541, layers.py:14-31
//   m_n = sum_{j in nbr(n)} h_j (.) W_{nj}                      nff/nn/modules.py:564-571, graphconv.py:43-53
//
// and every derivative of it that a force evaluation and the adjoint's force-vjp need
// (mdgrad_amd/nn/analytic.py: primal, forward-mode tangent along x_dot = w, and the reverse sweep of U_dot).
// Nothing edge-sized ever reaches HBM: the Gaussian basis is computed in registers as the MFMA A fragment, both
// Dense layers run on v_mfma_f32_16x16x4_f32 with W1 / W2 resident in LDS, the filter rows are multiplied with
// the gathered node rows in the accumulator layout and summed per atom in the wave.  The reverse sweeps
// RECOMPUTE the filter network instead of loading saved [E,G] / [E,F] tensors.
//
//   cfconv_fwd_kernel<GP,FT,TANGENT>   atom-centric (one wave per atom, 16 neighbour slots per MFMA tile):
//       m_n  = sum_s h[col_s] (.) W(d_s)
//       md_n = sum_s h[col_s] (.) Wd_s + hd[col_s] (.) W(d_s)          Wd = dW/dd * dd      (TANGENT)
//     The same kernel serves the reverse sweeps, because the aggregation is symmetric in the adjacency:
//     fed (mdb, mb) it returns (hdb, hb) = the adjoints of (hd, h).
//   cfconv_bwd_kernel<GP,FT,DUAL,THETA> edge-centric (16 undirected edges per tile): the adjoints of the filter
//     network.  With the adjoint rows  Wdb_e = mdb_i h_j + mdb_j h_i  (adjoint of Wd) and
//     Wb_e = mb_i h_j + mb_j h_i + mdb_i hd_j + mdb_j hd_i  (adjoint of W, DUAL):
//       dd_b[e] += d(Wdb . Wd)/d(dd)          d_b[e] += d(Wdb . Wd + Wb . W)/d(d)
//       gW1, gb1, gW2  += the parameter gradients of the same scalar                     (THETA)
//     dd_b is at the same time dU/dd of the plain reverse sweep (the adjoint of a tangent is the reverse-mode
//     adjoint of the primal), so the force falls out of the dual sweep: there is no separate reverse pass of U.
//
// MFMA operand conventions (v_mfma_f32_16x16x4_f32; lane = li + 16 lk, li < 16, lk < 4):
//   A: lane holds A[row li][k = lk]      B: lane holds B[k = lk][col li]      C: acc[r] = C[row 4 lk + r][col li]
// The contraction index may be any permutation as long as A and B use the same one; the kernels pick
// permutations that make the global gathers 16-byte vectors:
//   forward:  output column (nt, li) of the second layer <-> filter f = li*FT + nt, so a lane owns FT consecutive
//             filters of its 4 neighbour rows (one or two float4 per gathered row);
//   backward: k-step ks = 4 q + c of the [16 edges x F] x [F x G] product <-> filter f = 16 q + 4 lk + c
//             (float4 gathers, 64-byte segments per row and instruction); the weight-gradient products contract
//             over the 16 edges of the tile with edge = 4 lk + r, which is exactly the accumulator layout, so the
//             C registers of one product are the A / B operands of the next without passing through LDS.
// All sums have a fixed order (per-wave accumulators, ordered cross-wave / cross-block reduction): bitwise
// reproducible, no atomics.
#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.69314718055994531f;
constexpr float PAD_D = 1.0e4f;         // distance of an inert row: every Gaussian is exactly 0 there
// The gradient of the second filter layer's BIAS for free (reverse sweeps with parameter gradients, n_gaussians below the
// padded width GP): the padded column GP - 1 of the first layer has zero weights, so its pre-activation is its bias alone;
// with the bias ln(2 e - 1) its activation ssp(.) is exactly 1 on every edge, and  gW2[f][GP - 1] = sum_e Wb[e][f] * 1  is
// d/d b2[f] -- accumulated by the products that are there anyway.  Nothing else sees the column: W2, the transposed W1 and
// the tangent operand are zero there, so its adjoints and tangents vanish (s_b, a_b, g_b and the gW1 / gb1 / basis terms).
// (Without it the caller needs the neighbour sums of the node rows from the forward sweep and one more reduction over atoms.)
constexpr float B2COL_BIAS = 1.4899244f;

__host__ __device__ inline int s16m32(int n) {   // smallest s >= n with s % 32 == 16 (conflict-free B fetches)
    int s = (n + 31) / 32 * 32 - 16;
    if (s < n) s += 32;
    return s;
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// softplus(x) - ln 2 and sigmoid(x) from one exp2 (torch's softplus threshold 20)
__device__ __forceinline__ void ssp_sig(float x, float& s, float& sg) {
    const float ex = __builtin_amdgcn_exp2f(x * LOG2E);
    const bool big = x > 20.f;
    const float sp = __builtin_amdgcn_logf(1.0f + ex) * LN2;
    s = (big ? x : sp) - LN2;
    sg = big ? 1.0f : ex * __builtin_amdgcn_rcpf(1.0f + ex);
}

// Work split of a persistent grid over `n` items in units of `unit` consecutive items per workgroup step: XCD x owns the
// x-th contiguous eighth; inside it workgroup w of W starts at unit w and advances by W units.  [begin, end) and the
// step are in items.  Grids that are not a multiple of 8 fall back to one interleaved sweep over everything.
__device__ __forceinline__ void xcd_sweep(int n, int unit, int& begin, int& end, int& step) {
    const int nb = gridDim.x, b = blockIdx.x;
    if ((nb & 7) == 0 && nb >= 8) {
        const int W = nb >> 3, x = b & 7, w = b >> 3;
        const int units = (n + unit - 1) / unit;
        const int per = (units + 7) >> 3;                        // units per XCD
        const int u_lo = x * per, u_hi = min(units, u_lo + per);
        begin = (u_lo + w) * unit;
        end = min(n, u_hi * unit);
        step = W * unit;
        if (u_lo + w >= u_hi) begin = end;
    } else {
        begin = b * unit; end = n; step = nb * unit;
    }
}

struct FilterDev {
    const float *mu, *coef, *W1, *b1, *W2, *b2;
    int G, F;
    int RS;       // floats per row of the node feature matrices (= n_filters of the whole layer; F is this launch's
                  // chunk of <= 128 filters, W2 / b2 and the node pointers are offset to its first filter)
    int RS16;     // rows16 variants: bf16 elements per row of the GATHERED node matrices (h, hd, mb, mdb point at bf16 mirrors)
};

// ============================================================================================ forward
struct FwdArgs {
    FilterDev net;
    const float *d, *dd, *h, *hd;
    const int32_t *col, *eid, *cnt;
    int N, max_nbr;
    float *m, *md, *hsum, *hdsum;
    // STASH variants (round 6): the filter network's hidden activations per undirected edge, [E][GP] bf16 each, written once per
    // evaluation by filter_stash_kernel -- s = ssp(a) and its tangent sd = sigmoid(a) a_dot, exactly the bf16 operands the
    // recomputing kernel feeds its second Dense layer (zero rows for pairs beyond the cutoff)
    const unsigned short *st_s, *st_sd;
};

template <int FT>
__device__ __forceinline__ void load_row(const float* __restrict__ base, int row, int F, int RS, int li, bool ok,
                                         float (&out)[FT]) {
    // FT consecutive filters li*FT .. li*FT+FT-1 of one gathered row.  The load is UNCONDITIONAL: an inert slot reads row 0 and
    // a lane past the last filter column 0 -- valid, finite memory whose values meet a zero filter value / a zero mask in
    // every product they enter (masked slots: W = 0 and mr = 0; padded columns: zero weights, never stored) -- so a tile's
    // gathers are straight-line code that is in flight while the filter network is evaluated.  32-bit element offset from
    // the uniform base (the launcher checks n_atoms * row stride < 2^30).
    const unsigned off = (unsigned)(ok ? row : 0) * (unsigned)RS + (unsigned)(li * FT + FT <= F ? li * FT : 0);
    const float4* p = reinterpret_cast<const float4*>(base + off);
#pragma unroll
    for (int v = 0; v < FT / 4; ++v) {
        const float4 x = p[v];
        out[4 * v] = x.x; out[4 * v + 1] = x.y; out[4 * v + 2] = x.z; out[4 * v + 3] = x.w;
    }
}

// The same row out of a bf16 mirror (rows16 variants, FT = 8): ONE 16-byte load per gathered row instead of two, half the
// lines, and the row stays packed in four registers until its values are used; they widen exactly (bf16 -> f32 is a shift),
// every product and sum downstream is the f32 arithmetic of the f32-row kernels.
__device__ __forceinline__ uint4 load_row16(const unsigned short* __restrict__ base, int row, int F, int RS16, int li, bool ok) {
    const unsigned off = (unsigned)(ok ? row : 0) * (unsigned)RS16 + (unsigned)(li * 8 + 8 <= F ? li * 8 : 0);
    return *reinterpret_cast<const uint4*>(base + off);
}
__device__ __forceinline__ float row16_elem(const uint4& x, int e) {        // e: a compile-time constant after unrolling
    const unsigned w = (e >> 1) == 0 ? x.x : (e >> 1) == 1 ? x.y : (e >> 1) == 2 ? x.z : x.w;
    return __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16));
}

template <int FT>
__device__ __forceinline__ void store_row(float* __restrict__ base, int row, int F, int RS, int li, const float (&v)[FT]) {
    if (li * FT + FT <= F) {
        float4* p = reinterpret_cast<float4*>(base + ((unsigned)row * (unsigned)RS + (unsigned)(li * FT)));
#pragma unroll
        for (int q = 0; q < FT / 4; ++q) p[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
}

// (occupancy: the register allocator is asked for three waves per SIMD where round 2's build had them -- 168 registers for
//  the primal + tangent sweep; left alone it drifted to 224 = two waves after small edits, 253 -> 293 us at E = 459 k)
// SPLIT (round 6): few atoms -- a 192-atom water box is 48 workgroups of four atoms, each wave walking its atom's ~6 tiles of 16
// slots one after the other at ~190 dependent f32 MFMAs per tile (26 us per sweep on a chip that is 80 % idle).  Here a
// workgroup owns ONE atom and deals its tiles to the four waves (wave w: slots 16 w, 16 w + 64, ...); the four partial rows meet
// in LDS and are added in wave order, so the result does not depend on timing (it differs from the unsplit kernel's by the order
// of the f32 sums).  Chosen by the launcher from the atom count alone.
template <int GP, int FT, bool TANGENT, bool SUMS, bool SPLIT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GP == 32 ? (TANGENT ? (SUMS ? 2 : 3) : 4) : 1)))
void cfconv_fwd_kernel(const FwdArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int S1 = GP % 32 == 16 ? GP : GP + 16;          // s16m32(GP) for GP in {16,32,48,64}
    constexpr int FP = 16 * FT;
    constexpr int S2 = FP % 32 == 16 ? FP : FP + 16;
    constexpr int SA = GP + 2;
    constexpr int KS = GP / 4;
    float* w1s = sm;                        // [GP][S1]  B1[k][j] = W1[j][k]
    float* w2s = w1s + GP * S1;             // [GP][S2]  B2[k][nt*16+li] = W2[li*FT+nt][k]
    float* mus = w2s + GP * S2;             // [GP] centres, [GP] c log2e, [GP] 2c, [GP] b1, [FP] b2 (permuted)
    float* cfs = mus + GP;
    float* c2s = cfs + GP;
    float* b1s = c2s + GP;
    float* b2s = b1s + GP;
    float* h1s = b2s + FP;                  // [4 waves][16][SA]  (, [4][16][SA] tangent)
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave index: uniform)
    const int G = A.net.G, F = A.net.F, RS = A.net.RS;
    for (int t = tid; t < GP * GP; t += 256) {
        const int k = t / GP, j = t % GP;
        w1s[k * S1 + j] = (k < G && j < G) ? A.net.W1[j * G + k] : 0.f;
    }
    for (int t = tid; t < GP * FP; t += 256) {
        const int k = t % GP, c = t / GP;                       // consecutive threads: consecutive k of one W2 row
        const int f = (c & 15) * FT + (c >> 4);
        w2s[k * S2 + c] = (k < G && f < F) ? A.net.W2[(size_t)f * G + k] : 0.f;
    }
    for (int k = tid; k < GP; k += 256) {
        const float c = k < G ? A.net.coef[k] : 0.f;
        mus[k] = k < G ? A.net.mu[k] : 0.f;
        cfs[k] = c * LOG2E;
        c2s[k] = 2.f * c;
        b1s[k] = k < G ? A.net.b1[k] : 0.f;
    }
    for (int c = tid; c < FP; c += 256) {
        const int f = (c & 15) * FT + (c >> 4);
        b2s[c] = f < F ? A.net.b2[f] : 0.f;
    }
    __syncthreads();

    const int li = lane & 15, lk = lane >> 4;
    float* h1w = h1s + wid * 16 * SA;
    float* h1dw = h1s + (4 + wid) * 16 * SA;
    // XCD x (= blockIdx.x % 8: workgroups are dispatched round-robin over the XCDs) owns the x-th contiguous eighth of
    // the atoms, and its workgroups sweep that eighth TOGETHER (workgroup w takes atoms 4 (t W + w) + wave, t = 0, 1, ...):
    // at any time the XCD works on a window of consecutive -- i.e. spatially close -- atoms, so the node rows their
    // neighbours gather stay in its 4 MB L2 (a contiguous chunk per workgroup spreads the XCD over the whole replica:
    // 8 MB of rows in flight, measured gather-bound)
    const bool has_hd = A.hd != nullptr;
    int n_begin, n_end, n_step;
    if constexpr (SPLIT) { n_begin = blockIdx.x; n_end = A.N; n_step = gridDim.x; }
    else xcd_sweep(A.N, 4, n_begin, n_end, n_step);
    const int aw = SPLIT ? 0 : wid;                            // the wave's atom within the workgroup's unit
    const int tw0 = SPLIT ? 16 * wid : 0, tstep = SPLIT ? 64 : 16;   // its first tile and the distance to its next one
    // The slot indices (edge id of the lane's A-layout slot, neighbour ids of its four C-layout slots) of the NEXT tile --
    // the first tile of the wave's next atom after an atom's last one -- are requested before the current tile is worked
    // on: the distance and node-row gathers of a tile then start at once instead of behind a second round trip.
    // (the neighbour counts of the wave's next 64 atoms come in ONE vector load, lane k <-> the k-th atom ahead, and are read
    //  out with v_readlane: a count fetched per atom sat, as a loop-carried scalar, behind a round trip of its own per atom)
    int cnt_vec = 0, pf_e = 0, pf_j[4] = {0, 0, 0, 0};
    if (n_begin + aw < n_end) {
        const size_t rb = (size_t)(n_begin + aw) * A.max_nbr;
        pf_e = A.eid[rb + min(tw0 + li, A.max_nbr - 1)];
#pragma unroll
        for (int r = 0; r < 4; ++r) pf_j[r] = A.col[rb + min(tw0 + 4 * lk + r, A.max_nbr - 1)];
    }
    int kat = 0;
    for (int n = n_begin + aw; n < n_end; n += n_step, ++kat) {
        if ((kat & 63) == 0) {
            int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // the lane id, recomputed here:
            asm volatile("" : "+v"(ln));                          // nothing of this rare load's address arithmetic lives across the loops
            const long long na = (long long)n + (long long)ln * n_step;
            cnt_vec = na < n_end ? A.cnt[na] : 0;
        }
        const int cnt = __builtin_amdgcn_readlane(cnt_vec, kat & 63);
        float macc[FT], mdacc[FT], hs[FT], hds[FT];
#pragma unroll
        for (int v = 0; v < FT; ++v) macc[v] = mdacc[v] = hs[v] = hds[v] = 0.f;
        for (int t0 = tw0; t0 < cnt || t0 == tw0; t0 += tstep) {  // (a wave without a tile of this atom still hands the prefetch on)
            const int ea_raw = pf_e;
            int j_raw[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) j_raw[r] = pf_j[r];
            {
                const bool last = t0 + tstep >= cnt;
                const int n2 = last ? n + n_step : n, t2 = last ? tw0 : t0 + tstep;
                if (n2 < n_end) {
                    const size_t rb2 = (size_t)n2 * A.max_nbr;
                    pf_e = A.eid[rb2 + min(t2 + li, A.max_nbr - 1)];
#pragma unroll
                    for (int r = 0; r < 4; ++r) pf_j[r] = A.col[rb2 + min(t2 + 4 * lk + r, A.max_nbr - 1)];
                }
            }
            // ---- A-layout row (slot t0 + li): distance (and its tangent).  Both loads are unconditional (slot past the row's
            // end: entry 0) and independent of one another, so they leave together with the row gathers below -- guarded by
            // the validity of the distance, the tangent's load sat behind a full round trip of the distance's
            const bool vin = t0 + li < cnt;
            const int ea = vin ? ea_raw : 0;
            const float dload = A.d[ea];
            const float ddload = TANGENT ? A.dd[ea] : 0.f;
            const float draw = vin ? dload : -1.f;
            // a stored (Verlet) list may hold pairs that are beyond the cutoff now: mdg_edge_geom_masked marks them
            // d = -1 and they are skipped like the slots past the row's end
            const bool va = draw >= 0.f;
            // bit s (lanes 0..15) of the ballot: slot t0 + s is a real neighbour; mr[r] = 1 / 0 for the lane's four C-layout rows
            const unsigned nib = ((unsigned)__ballot(va) >> (4 * lk)) & 0xFu;
            bool mr[4];                                           // (lane masks: they live in scalar registers)
#pragma unroll
            for (int r = 0; r < 4; ++r) mr[r] = ((nib >> r) & 1u) != 0u;
            const float da = va ? draw : PAD_D;
            const float dda = (TANGENT && va) ? ddload : 0.f;
            // ---- C-layout rows (slots t0 + 4 lk + r): gathered node rows, FT consecutive filters per lane
            float hreg[4][FT], hdreg[4][FT];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = t0 + 4 * lk + r;
                const bool vc = s < cnt;                          // (the gathers do not wait for the distances: masked rows are
                const int j = vc ? j_raw[r] : 0;                  //  zeroed through the filter below)
                load_row<FT>(A.h, j, F, RS, li, vc, hreg[r]);
            }
            if (TANGENT) {
                if (has_hd) {                                     // (wave-uniform: the first block's node rows have no tangent)
#pragma unroll
                    for (int r = 0; r < 4; ++r) load_row<FT>(A.hd, j_raw[r], F, RS, li, t0 + 4 * lk + r < cnt, hdreg[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int v = 0; v < FT; ++v) hdreg[r][v] = 0.f;
                }
            }
            // ---- layer 1: Gaussians (and d/dd of them) in registers as A fragments
            float af[KS], adf[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int k = ks * 4 + lk;
                const float x = da - mus[k];
                const float g = __builtin_amdgcn_exp2f(cfs[k] * x * x);
                af[ks] = g;
                if (TANGENT) adf[ks] = g * (c2s[k] * x) * dda;
            }
#pragma unroll
            for (int nt = 0; nt < GP / 16; ++nt) {
                float bf[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) bf[ks] = w1s[(ks * 4 + lk) * S1 + nt * 16 + li];
                f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    acc = MFMA(af[ks], bf[ks], acc);
                    if (TANGENT) accd = MFMA(adf[ks], bf[ks], accd);
                }
                const int c = nt * 16 + li;
                const float bias = b1s[c];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s, sg;
                    ssp_sig(acc[r] + bias, s, sg);
                    h1w[(lk * 4 + r) * SA + c] = s;             // (padded columns: zero weights and bias -> ssp(0) = 0)
                    if (TANGENT) h1dw[(lk * 4 + r) * SA + c] = sg * accd[r];
                }
            }
            // (h1w / h1dw are private to the wave: program order + the LDS counter suffice, no barrier)
            // a masked slot's row of the second layer is zeroed at its A fragment (the lane's own row li) and its bias below:
            // W = 0 for it, whatever was gathered
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                af[ks] = va ? h1w[li * SA + ks * 4 + lk] : 0.f;
                if (TANGENT) adf[ks] = va ? h1dw[li * SA + ks * 4 + lk] : 0.f;
            }
            // ---- layer 2 + multiply with the gathered rows + sum over the 4 rows of the lane
#pragma unroll
            for (int nt = 0; nt < FT; ++nt) {
                float bf[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) bf[ks] = w2s[(ks * 4 + lk) * S2 + nt * 16 + li];
                f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    acc = MFMA(af[ks], bf[ks], acc);
                    if (TANGENT) accd = MFMA(adf[ks], bf[ks], accd);
                }
                const float bias = b2s[nt * 16 + li];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float W = acc[r] + (mr[r] ? bias : 0.f);
                    macc[nt] = fmaf(hreg[r][nt], W, macc[nt]);
                    if (SUMS) hs[nt] += mr[r] ? hreg[r][nt] : 0.f;
                    if (TANGENT) {
                        mdacc[nt] = fmaf(hreg[r][nt], accd[r], mdacc[nt]);
                        mdacc[nt] = fmaf(hdreg[r][nt], W, mdacc[nt]);
                        if (SUMS) hds[nt] += mr[r] ? hdreg[r][nt] : 0.f;
                    }
                }
            }
        }
        // ---- sum over the 4 row groups (lanes li, li+16, li+32, li+48) and store the atom's row
#pragma unroll
        for (int v = 0; v < FT; ++v) {
            macc[v] += __shfl_xor(macc[v], 16, 64); macc[v] += __shfl_xor(macc[v], 32, 64);
            if (TANGENT) { mdacc[v] += __shfl_xor(mdacc[v], 16, 64); mdacc[v] += __shfl_xor(mdacc[v], 32, 64); }
        }
        if (SUMS) {
#pragma unroll
            for (int v = 0; v < FT; ++v) {
                hs[v] += __shfl_xor(hs[v], 16, 64); hs[v] += __shfl_xor(hs[v], 32, 64);
                if (TANGENT) { hds[v] += __shfl_xor(hds[v], 16, 64); hds[v] += __shfl_xor(hds[v], 32, 64); }
            }
        }
        if constexpr (SPLIT) {
            // the waves' partial rows through their own (now idle) h1 areas: [quantity][filter v][li], 4 FT 16 <= 16 SA floats
            static_assert(4 * FT * 16 <= 16 * SA, "the partial rows fit the wave's layer-1 buffer");
            if (lk == 0) {
#pragma unroll
                for (int v = 0; v < FT; ++v) {
                    h1w[v * 16 + li] = macc[v];
                    if (TANGENT) h1w[(FT + v) * 16 + li] = mdacc[v];
                    if (SUMS) h1w[(2 * FT + v) * 16 + li] = hs[v];
                    if (SUMS && TANGENT) h1w[(3 * FT + v) * 16 + li] = hds[v];
                }
            }
            __syncthreads();
            if (wid == 0 && lk == 0) {
#pragma unroll
                for (int v = 0; v < FT; ++v) {
                    macc[v] = ((h1s[v * 16 + li] + h1s[16 * SA + v * 16 + li]) + h1s[2 * 16 * SA + v * 16 + li]) + h1s[3 * 16 * SA + v * 16 + li];
                    if (TANGENT) {
                        const int o = (FT + v) * 16 + li;
                        mdacc[v] = ((h1s[o] + h1s[16 * SA + o]) + h1s[2 * 16 * SA + o]) + h1s[3 * 16 * SA + o];
                    }
                    if (SUMS) {
                        const int o = (2 * FT + v) * 16 + li;
                        hs[v] = ((h1s[o] + h1s[16 * SA + o]) + h1s[2 * 16 * SA + o]) + h1s[3 * 16 * SA + o];
                    }
                    if (SUMS && TANGENT) {
                        const int o = (3 * FT + v) * 16 + li;
                        hds[v] = ((h1s[o] + h1s[16 * SA + o]) + h1s[2 * 16 * SA + o]) + h1s[3 * 16 * SA + o];
                    }
                }
            }
        }
        if (lk == 0 && (!SPLIT || wid == 0)) {
            store_row<FT>(A.m, n, F, RS, li, macc);
            if (TANGENT) store_row<FT>(A.md, n, F, RS, li, mdacc);
            if (SUMS) store_row<FT>(A.hsum, n, F, RS, li, hs);
            if (SUMS && TANGENT && A.hdsum) store_row<FT>(A.hdsum, n, F, RS, li, hds);
        }
        if constexpr (SPLIT) __syncthreads();                    // (the next atom's tiles overwrite the h1 areas)
    }
}

// ---------------------------------------------------------------------------------------------
// bf16-operand variant of the forward / tangent kernel (BASELINE config #5: "bf16 cfconv MFMA"):
// v_mfma_f32_16x16x32_bf16 -- K = 32 per instruction, so G <= 32 needs ONE MFMA per 16 x 16 tile of either Dense
// layer (20 per 16-slot tile for primal + tangent instead of 160 f32 ones).  Operands (Gaussians, W1, ssp output,
// W2) are rounded to bf16 (round-to-nearest-even); accumulation, biases, activation, the multiply with the gathered
// node rows and the per-atom sums stay fp32.  Operand layout: lane (li, lk) holds k = 32 ks + 8 lk + [0, 8).
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned int u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// two f32 -> two bf16 in ONE instruction (v_cvt_pk_bf16_f32, round to nearest even: the same bits as f2bf for finite
// values; the software rounding above costs four VALU operations per value and was a fifth of the bf16 kernels' issue
// slots); lo half = a, hi half = b
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2hw __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned int cvt_pk_bf16(float a, float b) {
    const f32x2v v = {a, b};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2hw));
}
__device__ __forceinline__ bf16x8 pack8(const float (&f)[8]) {
    const u32x4v u = {cvt_pk_bf16(f[0], f[1]), cvt_pk_bf16(f[2], f[3]), cvt_pk_bf16(f[4], f[5]), cvt_pk_bf16(f[6], f[7])};
    return __builtin_bit_cast(bf16x8, u);
}

// (three waves per SIMD for the primal + tangent sweep: 168 registers + 11 spilled dwords instead of 188 registers at two
//  waves -- the sweep waits on its gathers, not on issue slots: 155 -> 126 us at E = 459 k)
// HASHD: the node rows carry a tangent (false for the first interaction block, whose input rows do not depend on the
// positions: no loads, no products, no zero-filled registers for it).  BIASK: n_gaussians <= GP - 2, so the second Dense
// layer's bias rides in the two unused k columns of the product -- the shifted softplus is made to return exactly 1 there (a
// pseudo-bias of ln(2 e - 1) on zero weights) and W2's columns hold the bias split into a bf16 head and a bf16 remainder
// (2^-17 relative) -- instead of a masked add per filter value: the bias of a masked slot vanishes with the slot's zeroed
// operand row, and the epilogue is the three fused multiply-adds of the aggregation alone.
// (packed bf16 node rows leave room for one more wave per SIMD where the kernel carries no node tangent or the neighbour sums)
// STASH (round 6): the A operands of the second Dense layer come from the per-edge stash instead of being
// recomputed per directed slot -- no Gaussians, no first layer, no softplus (30 exp + 30 log + 30 rcp per slot and sweep, and
// every undirected edge from both ends): the sweep keeps the G -> F product on the MFMA and the multiply-sum epilogue.  The
// stash holds exactly the bf16 operands this kernel would have computed, so the results are bitwise the recomputing kernel's.
template <int GP, int FT, bool TANGENT, bool SUMS, bool HASHD, bool BIASK, bool R16 = false, bool STASH = false>          // GP in {32, 64}
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(GP == 32 ? (STASH ? (TANGENT ? 4 : 5)
                                                     : (TANGENT ? (SUMS ? (R16 ? 3 : 2) : ((R16 && !HASHD) ? 4 : 3)) : 4)) : 1)))
void cfconv_fwd_bf16_kernel(const FwdArgs A) {
    static_assert(TANGENT || !HASHD, "node tangents come with the tangent sweep");
    static_assert(!R16 || FT == 8, "bf16 node rows: layers of more than 64 filters");
    static_assert(!STASH || !SUMS, "the neighbour sums ride on the recomputing kernels");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int FP = 16 * FT;
    constexpr int KSB = GP + 8;                    // bf16 row stride (elements): 16-B aligned rows, conflict-free b128 reads
    constexpr int KB = GP / 32;
    float* mus = sm;                               // [GP] centres, [GP] c log2e, [GP] 2c, [GP] b1, [FP] b2 (permuted)
    float* cfs = mus + GP;
    float* c2s = cfs + GP;
    float* b1s = c2s + GP;
    float* b2s = b1s + GP;
    unsigned short* w1b = reinterpret_cast<unsigned short*>(b2s + FP);      // [GP rows j][KSB]   W1[j][k]
    unsigned short* w2b = w1b + GP * KSB;                                    // [FP rows c][KSB]   W2[f(c)][k]
    unsigned short* h1s = w2b + FP * KSB;                                    // [4 (+4) waves][16][KSB]
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave index: uniform)
    const int G = A.net.G, F = A.net.F, RS = A.net.RS;
    if constexpr (!STASH) {
    for (int t = tid; t < GP * GP; t += 256) {
        const int j = t / GP, k = t % GP;
        w1b[j * KSB + k] = (j < G && k < G) ? f2bf(A.net.W1[j * G + k]) : 0;
    }
    }
    for (int t = tid; t < FP * GP; t += 256) {
        const int c = t / GP, k = t % GP;
        const int f = (c & 15) * FT + (c >> 4);
        unsigned short wv = (f < F && k < G) ? f2bf(A.net.W2[(size_t)f * G + k]) : 0;
        if (BIASK && k >= GP - 2 && f < F) {                     // bias columns: head and remainder of b2[f]
            const float b = A.net.b2[f];
            const unsigned short hi = f2bf(b);
            wv = k == GP - 2 ? hi : f2bf(b - __uint_as_float((unsigned)hi << 16));
        }
        w2b[c * KSB + k] = wv;
    }
    for (int k = tid; k < GP; k += 256) {
        const float c = k < G ? A.net.coef[k] : 0.f;
        mus[k] = k < G ? A.net.mu[k] : 0.f;
        cfs[k] = c * LOG2E;
        c2s[k] = 2.f * c;
        b1s[k] = k < G ? A.net.b1[k] : ((BIASK && k >= GP - 2) ? 1.4899244f : 0.f);    // ssp(ln(2 e - 1)) = 1
    }
    for (int c = tid; c < FP; c += 256) {
        const int f = (c & 15) * FT + (c >> 4);
        b2s[c] = (!BIASK && f < F) ? A.net.b2[f] : 0.f;
    }
    __syncthreads();

    const int li = lane & 15, lk = lane >> 4;
    unsigned short* h1w = h1s + wid * 16 * KSB;
    unsigned short* h1dw = h1s + (4 + wid) * 16 * KSB;
    int n_begin, n_end, n_step;
    xcd_sweep(A.N, 4, n_begin, n_end, n_step);              // (see cfconv_fwd_kernel)
    // The slot indices (edge id of the lane's A-layout slot, neighbour ids of its four C-layout slots) of the NEXT tile --
    // the first tile of the wave's next atom after an atom's last one -- are requested before the current tile is worked
    // on: the distance and node-row gathers of a tile then start at once instead of behind a second round trip.
    // (the neighbour counts of the wave's next 64 atoms come in ONE vector load, lane k <-> the k-th atom ahead, and are read
    //  out with v_readlane: a count fetched per atom sat, as a loop-carried scalar, behind a round trip of its own per atom)
    int cnt_vec = 0, pf_e = 0, pf_j[4] = {0, 0, 0, 0};
    if (n_begin + wid < n_end) {
        const size_t rb = (size_t)(n_begin + wid) * A.max_nbr;
        pf_e = A.eid[rb + min(li, A.max_nbr - 1)];
#pragma unroll
        for (int r = 0; r < 4; ++r) pf_j[r] = A.col[rb + min(4 * lk + r, A.max_nbr - 1)];
    }
    int kat = 0;
    for (int n = n_begin + wid; n < n_end; n += n_step, ++kat) {
        if ((kat & 63) == 0) {
            int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // the lane id, recomputed here:
            asm volatile("" : "+v"(ln));                          // nothing of this rare load's address arithmetic lives across the loops
            const long long na = (long long)n + (long long)ln * n_step;
            cnt_vec = na < n_end ? A.cnt[na] : 0;
        }
        const int cnt = __builtin_amdgcn_readlane(cnt_vec, kat & 63);
        float macc[FT], mdacc[FT], hs[FT], hds[FT];
#pragma unroll
        for (int v = 0; v < FT; ++v) macc[v] = mdacc[v] = hs[v] = hds[v] = 0.f;
        for (int t0 = 0; t0 < cnt || t0 == 0; t0 += 16) {         // (an atom without neighbours still hands the prefetch on)
            const int ea_raw = pf_e;
            int j_raw[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) j_raw[r] = pf_j[r];
            {
                const bool last = t0 + 16 >= cnt;
                const int n2 = last ? n + n_step : n, t2 = last ? 0 : t0 + 16;
                if (n2 < n_end) {
                    const size_t rb2 = (size_t)n2 * A.max_nbr;
                    pf_e = A.eid[rb2 + min(t2 + li, A.max_nbr - 1)];
#pragma unroll
                    for (int r = 0; r < 4; ++r) pf_j[r] = A.col[rb2 + min(t2 + 4 * lk + r, A.max_nbr - 1)];
                }
            }
            const bool vin = t0 + li < cnt;
            const int ea = vin ? ea_raw : 0;
            // (STASH with the bias in the k columns: the stash row of a pair beyond the cutoff is zero -- bias columns included --
            //  so the sweep needs neither the distance nor its tangent; the slots past the row's end are zeroed by `vin`)
            constexpr bool NEED_D = !(STASH && BIASK);
            float dload = 0.f, ddload = 0.f;
            if constexpr (NEED_D) {
                dload = A.d[ea];                                  // (unconditional, with the tangent's: see cfconv_fwd_kernel)
                if constexpr (!STASH) ddload = TANGENT ? A.dd[ea] : 0.f;
            }
            bf16x8 sfr[STASH ? KB : 1], sdfr[(STASH && TANGENT) ? KB : 1];
            if constexpr (STASH) {                                // 16 bytes per lane and k-block: 64 contiguous bytes per edge and quantity
                const unsigned so = (unsigned)ea * (unsigned)GP + (unsigned)(lk * 8);
#pragma unroll
                for (int ks = 0; ks < KB; ++ks) {
                    sfr[ks] = *reinterpret_cast<const bf16x8*>(A.st_s + so + ks * 32);
                    if constexpr (TANGENT) sdfr[ks] = *reinterpret_cast<const bf16x8*>(A.st_sd + so + ks * 32);
                }
            }
            const float draw = NEED_D ? (vin ? dload : -1.f) : (vin ? 0.f : -1.f);
            // a stored (Verlet) list may hold pairs that are beyond the cutoff now: mdg_edge_geom_masked marks them
            // d = -1 and they are skipped like the slots past the row's end
            const bool va = draw >= 0.f;
            // bit s (lanes 0..15) of the ballot: slot t0 + s is a real neighbour; mr[r] = 1 / 0 for the lane's four C-layout rows
            const unsigned nib = ((unsigned)__ballot(va) >> (4 * lk)) & 0xFu;
            bool mr[4];                                           // (lane masks: they live in scalar registers)
#pragma unroll
            for (int r = 0; r < 4; ++r) mr[r] = ((nib >> r) & 1u) != 0u;
            const float da = va ? draw : PAD_D;
            const float dda = (TANGENT && va) ? ddload : 0.f;
            float hreg[R16 ? 1 : 4][FT], hdreg[R16 ? 1 : 4][FT];
            uint4 hraw[R16 ? 4 : 1], hdraw[R16 ? 4 : 1];          // R16: the packed rows
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = t0 + 4 * lk + r;
                const bool vc = s < cnt;                          // (the gathers do not wait for the distances: masked rows are
                const int j = vc ? j_raw[r] : 0;                  //  zeroed through the filter below)
                if constexpr (R16) hraw[r] = load_row16(reinterpret_cast<const unsigned short*>(A.h), j, F, A.net.RS16, li, vc);
                else load_row<FT>(A.h, j, F, RS, li, vc, hreg[r]);
            }
            if (HASHD) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (R16)
                        hdraw[r] = load_row16(reinterpret_cast<const unsigned short*>(A.hd), j_raw[r], F, A.net.RS16, li, t0 + 4 * lk + r < cnt);
                    else load_row<FT>(A.hd, j_raw[r], F, RS, li, t0 + 4 * lk + r < cnt, hdreg[r]);
                }
            }
            bf16x8 af[KB], adf[KB];
            if constexpr (STASH) {
#pragma unroll
                for (int ks = 0; ks < KB; ++ks) {
                    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
                    af[ks] = va ? sfr[ks] : zero8;
                    if (TANGENT) adf[ks] = va ? sdfr[ks] : zero8;
                }
            } else {
#pragma unroll
            for (int ks = 0; ks < KB; ++ks) {
                float gk[8], gdk[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int k = ks * 32 + lk * 8 + t;
                    const float x = da - mus[k];
                    const float g = __builtin_amdgcn_exp2f(cfs[k] * x * x);
                    gk[t] = g;
                    gdk[t] = TANGENT ? g * (c2s[k] * x) * dda : 0.f;
                }
                af[ks] = pack8(gk);
                if (TANGENT) adf[ks] = pack8(gdk);
            }
#pragma unroll
            for (int nt = 0; nt < GP / 16; ++nt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KB; ++ks) {
                    const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(&w1b[(nt * 16 + li) * KSB + ks * 32 + lk * 8]);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], bfr, acc, 0, 0, 0);
                    if (TANGENT) accd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(adf[ks], bfr, accd, 0, 0, 0);
                }
                const int c = nt * 16 + li;
                const float bias = b1s[c];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s, sg;
                    ssp_sig(acc[r] + bias, s, sg);
                    h1w[(lk * 4 + r) * KSB + c] = (unsigned short)cvt_pk_bf16(s, 0.f);
                    if (TANGENT) h1dw[(lk * 4 + r) * KSB + c] = (unsigned short)cvt_pk_bf16(sg * accd[r], 0.f);
                }
            }
#pragma unroll
            for (int ks = 0; ks < KB; ++ks) {
                const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};      // (masked slot: see cfconv_fwd_kernel)
                af[ks] = va ? *reinterpret_cast<const bf16x8*>(&h1w[li * KSB + ks * 32 + lk * 8]) : zero8;
                if (TANGENT) adf[ks] = va ? *reinterpret_cast<const bf16x8*>(&h1dw[li * KSB + ks * 32 + lk * 8]) : zero8;
            }
            }
#pragma unroll
            for (int nt = 0; nt < FT; ++nt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KB; ++ks) {
                    const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(&w2b[(nt * 16 + li) * KSB + ks * 32 + lk * 8]);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], bfr, acc, 0, 0, 0);
                    if (TANGENT) accd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(adf[ks], bfr, accd, 0, 0, 0);
                }
                const float bias = BIASK ? 0.f : b2s[nt * 16 + li];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float W = BIASK ? acc[r] : acc[r] + (mr[r] ? bias : 0.f);
                    float hv, hdv = 0.f;
                    if constexpr (R16) { hv = row16_elem(hraw[r], nt); if (HASHD) hdv = row16_elem(hdraw[r], nt); }
                    else { hv = hreg[r][nt]; if (HASHD) hdv = hdreg[r][nt]; }
                    macc[nt] = fmaf(hv, W, macc[nt]);
                    if (SUMS) hs[nt] += mr[r] ? hv : 0.f;
                    if (TANGENT) {
                        mdacc[nt] = fmaf(hv, accd[r], mdacc[nt]);
                        if (HASHD) mdacc[nt] = fmaf(hdv, W, mdacc[nt]);
                        if (SUMS && HASHD) hds[nt] += mr[r] ? hdv : 0.f;
                    }
                }
            }
        }
#pragma unroll
        for (int v = 0; v < FT; ++v) {
            macc[v] += __shfl_xor(macc[v], 16, 64); macc[v] += __shfl_xor(macc[v], 32, 64);
            if (TANGENT) { mdacc[v] += __shfl_xor(mdacc[v], 16, 64); mdacc[v] += __shfl_xor(mdacc[v], 32, 64); }
        }
        if (SUMS) {
#pragma unroll
            for (int v = 0; v < FT; ++v) {
                hs[v] += __shfl_xor(hs[v], 16, 64); hs[v] += __shfl_xor(hs[v], 32, 64);
                if (HASHD) { hds[v] += __shfl_xor(hds[v], 16, 64); hds[v] += __shfl_xor(hds[v], 32, 64); }
            }
        }
        if (lk == 0) {
            store_row<FT>(A.m, n, F, RS, li, macc);
            if (TANGENT) store_row<FT>(A.md, n, F, RS, li, mdacc);
            if (SUMS) store_row<FT>(A.hsum, n, F, RS, li, hs);
            if (SUMS && HASHD && A.hdsum) store_row<FT>(A.hdsum, n, F, RS, li, hds);
        }
    }
}

template <int GP, int FT>
size_t fwd_bf16_lds_bytes(bool tangent) {
    constexpr int FP = 16 * FT, KSB = GP + 8;
    return sizeof(float) * (4 * GP + FP) + sizeof(unsigned short) * ((size_t)GP * KSB + (size_t)FP * KSB + (tangent ? 8 : 4) * 16 * KSB);
}

// ---------------------------------------------------------------------------------------------
// X6 (round 6): the f32 sweep on the bf16 matrix pipe at f32 accuracy.  Every operand of the two Dense layers -- Gaussians,
// W1, the softplus output, W2 -- is cut into THREE bf16 pieces (8 + 8 + 8 significant bits: the pieces add up to the f32 value
// exactly) and a product is the six piece products that matter,
//     a w  =  a1 w1 + (a1 w2 + a2 w1) + (a2 w2 + a1 w3 + a3 w1)  (+ terms below 2^-25 of it),
// each exact in the f32 accumulator.  tools/micro/split_mfma.hip: max error 1.8e-7 of sum |terms| at K = 128 (rms 1.8e-8)
// against 1.9e-7 (2.4e-8) for v_mfma_f32_16x16x4_f32 -- and 6 x 16 cycles per 32 k instead of 8 x 32.  The f32 sweep is bound
// by exactly that issue time (mfma_busy 0.50-0.58; ~190 dependent matrix instructions per 16-slot tile).  Layout and slot
// logic are cfconv_fwd_kernel's (f32 node rows, lane li owns FT consecutive filters, biases added in f32); operand layout is
// the bf16 kernel's (lane (li, lk): k = 8 lk + [0, 8)).  n_gaussians <= 32.  SPLIT as in cfconv_fwd_kernel.
__device__ __forceinline__ float bf_up(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ void split3(float w, unsigned short& p1, unsigned short& p2, unsigned short& p3) {
    p1 = f2bf(w);
    const float r1 = w - bf_up(p1);
    p2 = f2bf(r1);
    p3 = f2bf(r1 - bf_up(p2));
}
// the three pieces of two values, packed (low half = a)
__device__ __forceinline__ void split2x3(float a, float b, unsigned& u1, unsigned& u2, unsigned& u3) {
    u1 = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(u1 << 16), rb = b - __uint_as_float(u1 & 0xffff0000u);
    u2 = cvt_pk_bf16(ra, rb);
    u3 = cvt_pk_bf16(ra - __uint_as_float(u2 << 16), rb - __uint_as_float(u2 & 0xffff0000u));
}
__device__ __forceinline__ void split8x3(const float (&f)[8], bf16x8 (&p)[3]) {
    u32x4v u1, u2, u3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = f[2 * i], b = f[2 * i + 1];
        const unsigned h = cvt_pk_bf16(a, b);
        const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
        const unsigned m = cvt_pk_bf16(ra, rb);
        const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
        u1[i] = h; u2[i] = m; u3[i] = cvt_pk_bf16(sa, sb);
    }
    p[0] = __builtin_bit_cast(bf16x8, u1); p[1] = __builtin_bit_cast(bf16x8, u2); p[2] = __builtin_bit_cast(bf16x8, u3);
}
// smallest pieces first
__device__ __forceinline__ f32x4 six(const bf16x8 (&a)[3], const bf16x8 (&b)[3]) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
    return acc;
}

template <int FT>
size_t fwd_x6_lds_bytes(bool tangent) {
    constexpr int GP = 32, FP = 16 * FT, KSB = GP + 8;
    return sizeof(float) * (4 * GP + FP) + sizeof(unsigned short) * 3 * ((size_t)GP * KSB + (size_t)FP * KSB + (tangent ? 8 : 4) * 16 * KSB);
}

template <int FT, bool TANGENT, bool SUMS, bool HASHD, bool SPLIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
void cfconv_fwd_x6_kernel(const FwdArgs A) {
    static_assert(TANGENT || !HASHD, "node tangents come with the tangent sweep");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int GP = 32, FP = 16 * FT, KSB = GP + 8, PL = 16 * KSB;      // PL: elements of one [16][KSB] piece plane
    float* mus = sm;                               // [GP] centres, [GP] c log2e, [GP] 2c, [GP] b1, [FP] b2 (permuted)
    float* cfs = mus + GP;
    float* c2s = cfs + GP;
    float* b1s = c2s + GP;
    float* b2s = b1s + GP;
    unsigned short* w1b = reinterpret_cast<unsigned short*>(b2s + FP);      // [3 pieces][GP rows j][KSB]   W1[j][k]
    unsigned short* w2b = w1b + 3 * GP * KSB;                               // [3][FP rows c][KSB]          W2[f(c)][k]
    unsigned short* h1s = w2b + 3 * FP * KSB;                               // [4 (+4) waves][3][16][KSB]
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = A.net.G, F = A.net.F, RS = A.net.RS;
    for (int t = tid; t < GP * GP; t += 256) {
        const int j = t / GP, k = t % GP;
        unsigned short p1, p2, p3;
        split3((j < G && k < G) ? A.net.W1[j * G + k] : 0.f, p1, p2, p3);
        w1b[j * KSB + k] = p1; w1b[GP * KSB + j * KSB + k] = p2; w1b[2 * GP * KSB + j * KSB + k] = p3;
    }
    for (int t = tid; t < FP * GP; t += 256) {
        const int c = t / GP, k = t % GP;
        const int f = (c & 15) * FT + (c >> 4);
        unsigned short p1, p2, p3;
        split3((f < F && k < G) ? A.net.W2[(size_t)f * G + k] : 0.f, p1, p2, p3);
        w2b[c * KSB + k] = p1; w2b[FP * KSB + c * KSB + k] = p2; w2b[2 * FP * KSB + c * KSB + k] = p3;
    }
    for (int k = tid; k < GP; k += 256) {
        const float c = k < G ? A.net.coef[k] : 0.f;
        mus[k] = k < G ? A.net.mu[k] : 0.f;
        cfs[k] = c * LOG2E;
        c2s[k] = 2.f * c;
        b1s[k] = k < G ? A.net.b1[k] : 0.f;
    }
    for (int c = tid; c < FP; c += 256) {
        const int f = (c & 15) * FT + (c >> 4);
        b2s[c] = f < F ? A.net.b2[f] : 0.f;
    }
    __syncthreads();

    const int li = lane & 15, lk = lane >> 4;
    unsigned short* h1w = h1s + wid * 3 * PL;
    unsigned short* h1dw = h1s + (4 + wid) * 3 * PL;
    int n_begin, n_end, n_step;
    if constexpr (SPLIT) { n_begin = blockIdx.x; n_end = A.N; n_step = gridDim.x; }
    else xcd_sweep(A.N, 4, n_begin, n_end, n_step);            // (see cfconv_fwd_kernel)
    const int aw = SPLIT ? 0 : wid;
    const int tw0 = SPLIT ? 16 * wid : 0, tstep = SPLIT ? 64 : 16;
    int cnt_vec = 0, pf_e = 0, pf_j[4] = {0, 0, 0, 0};
    if (n_begin + aw < n_end) {
        const size_t rb = (size_t)(n_begin + aw) * A.max_nbr;
        pf_e = A.eid[rb + min(tw0 + li, A.max_nbr - 1)];
#pragma unroll
        for (int r = 0; r < 4; ++r) pf_j[r] = A.col[rb + min(tw0 + 4 * lk + r, A.max_nbr - 1)];
    }
    int kat = 0;
    for (int n = n_begin + aw; n < n_end; n += n_step, ++kat) {
        if ((kat & 63) == 0) {
            int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            asm volatile("" : "+v"(ln));
            const long long na = (long long)n + (long long)ln * n_step;
            cnt_vec = na < n_end ? A.cnt[na] : 0;
        }
        const int cnt = __builtin_amdgcn_readlane(cnt_vec, kat & 63);
        float macc[FT], mdacc[FT], hs[FT], hds[FT];
#pragma unroll
        for (int v = 0; v < FT; ++v) macc[v] = mdacc[v] = hs[v] = hds[v] = 0.f;
        for (int t0 = tw0; t0 < cnt || t0 == tw0; t0 += tstep) {
            const int ea_raw = pf_e;
            int j_raw[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) j_raw[r] = pf_j[r];
            {
                const bool last = t0 + tstep >= cnt;
                const int n2 = last ? n + n_step : n, t2 = last ? tw0 : t0 + tstep;
                if (n2 < n_end) {
                    const size_t rb2 = (size_t)n2 * A.max_nbr;
                    pf_e = A.eid[rb2 + min(t2 + li, A.max_nbr - 1)];
#pragma unroll
                    for (int r = 0; r < 4; ++r) pf_j[r] = A.col[rb2 + min(t2 + 4 * lk + r, A.max_nbr - 1)];
                }
            }
            const bool vin = t0 + li < cnt;
            const int ea = vin ? ea_raw : 0;
            const float dload = A.d[ea];
            const float ddload = TANGENT ? A.dd[ea] : 0.f;
            const float draw = vin ? dload : -1.f;
            const bool va = draw >= 0.f;                          // (a stored list's pairs beyond the cutoff carry d = -1)
            const unsigned nib = ((unsigned)__ballot(va) >> (4 * lk)) & 0xFu;
            bool mr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) mr[r] = ((nib >> r) & 1u) != 0u;
            const float da = va ? draw : PAD_D;
            const float dda = (TANGENT && va) ? ddload : 0.f;
            float hreg[4][FT], hdreg[HASHD ? 4 : 1][FT];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool vc = t0 + 4 * lk + r < cnt;
                load_row<FT>(A.h, vc ? j_raw[r] : 0, F, RS, li, vc, hreg[r]);
            }
            if constexpr (HASHD) {
#pragma unroll
                for (int r = 0; r < 4; ++r) load_row<FT>(A.hd, j_raw[r], F, RS, li, t0 + 4 * lk + r < cnt, hdreg[r]);
            }
            // ---- layer 1: Gaussians (and d/dd of them), three pieces each
            bf16x8 ga[3], gda[3];
            {
                float gk[8], gdk[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int k = lk * 8 + t;
                    const float x = da - mus[k];
                    const float g = __builtin_amdgcn_exp2f(cfs[k] * x * x);
                    gk[t] = g;
                    gdk[t] = TANGENT ? g * (c2s[k] * x) * dda : 0.f;
                }
                split8x3(gk, ga);
                if (TANGENT) split8x3(gdk, gda);
            }
#pragma unroll
            for (int nt = 0; nt < GP / 16; ++nt) {
                bf16x8 b[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8*>(&w1b[p * GP * KSB + (nt * 16 + li) * KSB + lk * 8]);
                const f32x4 acc = six(ga, b);
                f32x4 accd = {0.f, 0.f, 0.f, 0.f};
                if (TANGENT) accd = six(gda, b);
                const int c = nt * 16 + li;
                const float bias = b1s[c];
                float sv[4], sdv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float sg;
                    ssp_sig(acc[r] + bias, sv[r], sg);
                    sdv[r] = TANGENT ? sg * accd[r] : 0.f;
                }
                // three pieces of two values at a time (v_cvt_pk_bf16_f32: the rounding of f2bf in one instruction)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const int o0 = (lk * 4 + r) * KSB + c, o1 = o0 + KSB;
                    unsigned u1, u2, u3;
                    split2x3(sv[r], sv[r + 1], u1, u2, u3);
                    h1w[o0] = (unsigned short)u1; h1w[o1] = (unsigned short)(u1 >> 16);
                    h1w[PL + o0] = (unsigned short)u2; h1w[PL + o1] = (unsigned short)(u2 >> 16);
                    h1w[2 * PL + o0] = (unsigned short)u3; h1w[2 * PL + o1] = (unsigned short)(u3 >> 16);
                    if (TANGENT) {
                        split2x3(sdv[r], sdv[r + 1], u1, u2, u3);
                        h1dw[o0] = (unsigned short)u1; h1dw[o1] = (unsigned short)(u1 >> 16);
                        h1dw[PL + o0] = (unsigned short)u2; h1dw[PL + o1] = (unsigned short)(u2 >> 16);
                        h1dw[2 * PL + o0] = (unsigned short)u3; h1dw[2 * PL + o1] = (unsigned short)(u3 >> 16);
                    }
                }
            }
            // (h1w / h1dw are private to the wave: program order + the LDS counter suffice, no barrier)
            bf16x8 sa[3], sda[3];
            {
                const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    sa[p] = va ? *reinterpret_cast<const bf16x8*>(&h1w[p * PL + li * KSB + lk * 8]) : zero8;
                    if (TANGENT) sda[p] = va ? *reinterpret_cast<const bf16x8*>(&h1dw[p * PL + li * KSB + lk * 8]) : zero8;
                }
            }
            // ---- layer 2 + multiply with the gathered rows + sum over the 4 rows of the lane
#pragma unroll
            for (int nt = 0; nt < FT; ++nt) {
                bf16x8 b[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8*>(&w2b[p * FP * KSB + (nt * 16 + li) * KSB + lk * 8]);
                const f32x4 acc = six(sa, b);
                f32x4 accd = {0.f, 0.f, 0.f, 0.f};
                if (TANGENT) accd = six(sda, b);
                const float bias = b2s[nt * 16 + li];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float W = acc[r] + (mr[r] ? bias : 0.f);
                    const float hv = hreg[r][nt];
                    macc[nt] = fmaf(hv, W, macc[nt]);
                    if (SUMS) hs[nt] += mr[r] ? hv : 0.f;
                    if (TANGENT) {
                        mdacc[nt] = fmaf(hv, accd[r], mdacc[nt]);
                        if constexpr (HASHD) {
                            mdacc[nt] = fmaf(hdreg[r][nt], W, mdacc[nt]);
                            if (SUMS) hds[nt] += mr[r] ? hdreg[r][nt] : 0.f;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int v = 0; v < FT; ++v) {
            macc[v] += __shfl_xor(macc[v], 16, 64); macc[v] += __shfl_xor(macc[v], 32, 64);
            if (TANGENT) { mdacc[v] += __shfl_xor(mdacc[v], 16, 64); mdacc[v] += __shfl_xor(mdacc[v], 32, 64); }
        }
        if (SUMS) {
#pragma unroll
            for (int v = 0; v < FT; ++v) {
                hs[v] += __shfl_xor(hs[v], 16, 64); hs[v] += __shfl_xor(hs[v], 32, 64);
                if (HASHD) { hds[v] += __shfl_xor(hds[v], 16, 64); hds[v] += __shfl_xor(hds[v], 32, 64); }
            }
        }
        if constexpr (SPLIT) {
            // the waves' partial rows through their own (now idle) piece planes: [quantity][filter v][li] floats
            static_assert(4 * FT * 16 * sizeof(float) <= 3 * PL * sizeof(unsigned short), "the partial rows fit the wave's planes");
            constexpr int WS = 3 * PL / 2;                        // a wave's planes, in floats
            float* part = reinterpret_cast<float*>(h1s);
            if (lk == 0) {
#pragma unroll
                for (int v = 0; v < FT; ++v) {
                    part[wid * WS + v * 16 + li] = macc[v];
                    if (TANGENT) part[wid * WS + (FT + v) * 16 + li] = mdacc[v];
                    if (SUMS) part[wid * WS + (2 * FT + v) * 16 + li] = hs[v];
                    if (SUMS && HASHD) part[wid * WS + (3 * FT + v) * 16 + li] = hds[v];
                }
            }
            __syncthreads();
            if (wid == 0 && lk == 0) {
#pragma unroll
                for (int v = 0; v < FT; ++v) {
                    const int o0 = v * 16 + li, o1 = (FT + v) * 16 + li, o2 = (2 * FT + v) * 16 + li, o3 = (3 * FT + v) * 16 + li;
                    macc[v] = ((part[o0] + part[WS + o0]) + part[2 * WS + o0]) + part[3 * WS + o0];
                    if (TANGENT) mdacc[v] = ((part[o1] + part[WS + o1]) + part[2 * WS + o1]) + part[3 * WS + o1];
                    if (SUMS) hs[v] = ((part[o2] + part[WS + o2]) + part[2 * WS + o2]) + part[3 * WS + o2];
                    if (SUMS && HASHD) hds[v] = ((part[o3] + part[WS + o3]) + part[2 * WS + o3]) + part[3 * WS + o3];
                }
            }
        }
        if (lk == 0 && (!SPLIT || wid == 0)) {
            store_row<FT>(A.m, n, F, RS, li, macc);
            if (TANGENT) store_row<FT>(A.md, n, F, RS, li, mdacc);
            if (SUMS) store_row<FT>(A.hsum, n, F, RS, li, hs);
            if (SUMS && HASHD && A.hdsum) store_row<FT>(A.hdsum, n, F, RS, li, hds);
        }
        if constexpr (SPLIT) __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------
// The stash producer (round 6): the first Dense layer of the filter network once per undirected edge and evaluation --
// Gaussians, a = g W1^T + b1, s = ssp(a), sd = sigmoid(a) a_dot -- with the ARITHMETIC OF cfconv_fwd_bf16_kernel (same bf16
// operands, same MFMA, same rounding of the outputs), written as [E][GP] bf16 rows in the A-operand order of the second
// layer's MFMA: 64 contiguous bytes per edge and quantity at GP = 32.  Pairs a stored list holds beyond the cutoff
// (d = -1) and padding rows get zero rows: a consumer then needs no distance.  nff/nn/modules.py:531-541 (layers 0-2 of
// message_edge_filter).
struct StashArgs {
    FilterDev net;
    const float *d, *dd;
    long long E;
    const int32_t* n_valid;
    unsigned short *st_s, *st_sd;
};

template <int GP, bool TANGENT, bool BIASK>
__global__ __launch_bounds__(256) void filter_stash_kernel(const StashArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int KSB = GP + 8, KB = GP / 32;
    float* mus = sm;
    float* cfs = mus + GP;
    float* c2s = cfs + GP;
    float* b1s = c2s + GP;
    unsigned short* w1b = reinterpret_cast<unsigned short*>(b1s + GP);        // [GP rows j][KSB]   W1[j][k]
    unsigned short* h1s = w1b + GP * KSB;                                      // [4 (+4) waves][16][KSB]
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = A.net.G;
    for (int t = tid; t < GP * GP; t += 256) {
        const int j = t / GP, k = t % GP;
        w1b[j * KSB + k] = (j < G && k < G) ? f2bf(A.net.W1[j * G + k]) : 0;
    }
    for (int k = tid; k < GP; k += 256) {
        const float c = k < G ? A.net.coef[k] : 0.f;
        mus[k] = k < G ? A.net.mu[k] : 0.f;
        cfs[k] = c * LOG2E;
        c2s[k] = 2.f * c;
        b1s[k] = k < G ? A.net.b1[k] : ((BIASK && k >= GP - 2) ? 1.4899244f : 0.f);    // ssp(ln(2 e - 1)) = 1: the bias columns
    }
    __syncthreads();
    const int li = lane & 15, lk = lane >> 4;
    unsigned short* h1w = h1s + wid * 16 * KSB;
    unsigned short* h1dw = h1s + (4 + wid) * 16 * KSB;
    const long long nrows = A.n_valid ? min((long long)*A.n_valid, A.E) : A.E;
    const long long ntiles = (A.E + 63) / 64;                    // (every row of the stash is written: padding rows as zeros)
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long e = tile * 64 + wid * 16 + li;
        const bool in = e < nrows;
        const float dload = in ? A.d[e] : -1.f;
        const float ddload = (TANGENT && in) ? A.dd[e] : 0.f;
        const bool va = dload >= 0.f;
        const float da = va ? dload : PAD_D;
        const float dda = (TANGENT && va) ? ddload : 0.f;
        bf16x8 af[KB], adf[KB];
#pragma unroll
        for (int ks = 0; ks < KB; ++ks) {
            float gk[8], gdk[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int k = ks * 32 + lk * 8 + t;
                const float x = da - mus[k];
                const float g = __builtin_amdgcn_exp2f(cfs[k] * x * x);
                gk[t] = g;
                gdk[t] = TANGENT ? g * (c2s[k] * x) * dda : 0.f;
            }
            af[ks] = pack8(gk);
            if (TANGENT) adf[ks] = pack8(gdk);
        }
#pragma unroll
        for (int nt = 0; nt < GP / 16; ++nt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KB; ++ks) {
                const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(&w1b[(nt * 16 + li) * KSB + ks * 32 + lk * 8]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], bfr, acc, 0, 0, 0);
                if (TANGENT) accd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(adf[ks], bfr, accd, 0, 0, 0);
            }
            const int c = nt * 16 + li;
            const float bias = b1s[c];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sv, sg;
                ssp_sig(acc[r] + bias, sv, sg);
                h1w[(lk * 4 + r) * KSB + c] = (unsigned short)cvt_pk_bf16(sv, 0.f);
                if (TANGENT) h1dw[(lk * 4 + r) * KSB + c] = (unsigned short)cvt_pk_bf16(sg * accd[r], 0.f);
            }
        }
        // (h1w / h1dw are private to the wave: program order + the LDS counter suffice, no barrier)
        if (e < A.E) {
            const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KB; ++ks) {
                const size_t o = (size_t)e * GP + ks * 32 + lk * 8;
                *reinterpret_cast<bf16x8*>(A.st_s + o) = va ? *reinterpret_cast<const bf16x8*>(&h1w[li * KSB + ks * 32 + lk * 8]) : zero8;
                if (TANGENT)
                    *reinterpret_cast<bf16x8*>(A.st_sd + o) = va ? *reinterpret_cast<const bf16x8*>(&h1dw[li * KSB + ks * 32 + lk * 8]) : zero8;
            }
        }
    }
}

template <int GP>
size_t stash_lds_bytes(bool tangent) {
    constexpr int KSB = GP + 8;
    return sizeof(float) * 4 * GP + sizeof(unsigned short) * ((size_t)GP * KSB + (tangent ? 8 : 4) * 16 * KSB);
}

template <int GP, int FT>
size_t fwd_lds_bytes(bool tangent) {
    constexpr int S1 = GP % 32 == 16 ? GP : GP + 16;
    constexpr int FP = 16 * FT;
    constexpr int S2 = FP % 32 == 16 ? FP : FP + 16;
    return sizeof(float) * ((size_t)GP * S1 + (size_t)GP * S2 + 4 * GP + FP + (tangent ? 8 : 4) * 16 * (GP + 2));
}

// ============================================================================================ backward
struct BwdArgs {
    FilterDev net;
    const float *d, *dd;
    const int64_t* nbr;
    long long E;
    const float *h, *hd, *mb, *mdb;
    float *d_b, *dd_b;
    float* part;              // THETA: [gridDim.x][GP*GP + GP + FP*GP] per-workgroup partial gradients
    const int32_t* n_valid;   // fixed-capacity lists: device count of real rows (rows beyond are padding), or null
};

// ---- the gathers of one 16-edge tile, shared by the f32 and the bf16 sweep ------------------------------------------
// Lane (li, lk) owns edge e0 + li and, per k-block q, filters 16 q + 4 lk + [0, 4) of it.
struct EdgeIdx {
    int i, j;            // atoms of the lane's edge (-1: padding row / past the end)
    float d, dd;         // its distance (-1: padding, or a pair of a stored list that is beyond the cutoff now) and tangent
};

// nbr, d and dd are indexed by the edge itself, so the three loads are independent: ONE round trip, requested a tile ahead
template <bool DUAL>
__device__ __forceinline__ EdgeIdx load_edge_idx(const BwdArgs& A, long long e) {
    EdgeIdx x{-1, -1, -1.f, 0.f};
    if (e < A.E) {
        const longlong2 ij = *reinterpret_cast<const longlong2*>(A.nbr + 2 * e);
        x.i = (int)ij.x; x.j = (int)ij.y;
        x.d = A.d[e];
        if (DUAL) x.dd = A.dd[e];
    }
    if (x.i < 0) x.d = -1.f;
    return x;
}

// Adjoint rows of the filter output:  Wdb = mdb_i h_j + mdb_j h_i,   Wb = mb_i h_j + mb_j h_i (+ mdb_i hd_j + mdb_j hd_i).
// Every load is UNCONDITIONAL -- the rows of an inert edge and the columns past F are clamped to row 0 / column 0 and the
// products selected to 0 afterwards -- and the k-blocks are gathered in batches of QB: all loads of a batch are issued back to
// back (a scheduling fence keeps the products behind them), so a tile costs FT / QB round trips with QB x 4..8 rows in flight
// per lane.  (Round 3 guarded each k-block's loads with the lane's validity: the divergent branches pinned every block's
// products behind its own loads, 8-16 dependent round trips per tile -- the sweep waited on memory 65 % of the time with one
// wave per SIMD; profiles/pmc_schnet4096.json.)
template <int FT, int QB, bool DUAL, bool HASHD>
__device__ __forceinline__ void gather_adjoint_rows(const BwdArgs& A, int ia, int ja, bool va, int lk, int F, int RS,
                                                    float (&wdb)[4 * FT], float (&wb)[DUAL ? 4 * FT : 1]) {
    const long long ri = (long long)(va ? ia : 0) * RS + 4 * lk, rj = (long long)(va ? ja : 0) * RS + 4 * lk;
    const float* __restrict__ hI = A.h + ri;
    const float* __restrict__ hJ = A.h + rj;
    const float* __restrict__ pI = A.mdb + ri;
    const float* __restrict__ pJ = A.mdb + rj;
    const float* __restrict__ bI = DUAL ? A.mb + ri : nullptr;
    const float* __restrict__ bJ = DUAL ? A.mb + rj : nullptr;
    const float* __restrict__ tI = HASHD ? A.hd + ri : nullptr;
    const float* __restrict__ tJ = HASHD ? A.hd + rj : nullptr;
#pragma unroll
    for (int q0 = 0; q0 < FT; q0 += QB) {
        float4 hi[QB], hj[QB], pi[QB], pj[QB], bi[DUAL ? QB : 1], bj[DUAL ? QB : 1], ti[HASHD ? QB : 1], tj[HASHD ? QB : 1];
        // (array by array: the 64-byte pieces two neighbouring k-blocks take of a row are the halves of one 128-byte line --
        //  requested back to back, the second is served by the fill the first started)
        int fc[QB];
#pragma unroll
        for (int u = 0; u < QB; ++u) fc[u] = 16 * (q0 + u) + 4 * lk + 4 <= F ? 16 * (q0 + u) : 0;
#pragma unroll
        for (int u = 0; u < QB; ++u) hj[u] = *reinterpret_cast<const float4*>(hJ + fc[u]);
#pragma unroll
        for (int u = 0; u < QB; ++u) pj[u] = *reinterpret_cast<const float4*>(pJ + fc[u]);
        if (DUAL) {
#pragma unroll
            for (int u = 0; u < QB; ++u) bj[u] = *reinterpret_cast<const float4*>(bJ + fc[u]);
        }
        if (HASHD) {
#pragma unroll
            for (int u = 0; u < QB; ++u) tj[u] = *reinterpret_cast<const float4*>(tJ + fc[u]);
        }
#pragma unroll
        for (int u = 0; u < QB; ++u) hi[u] = *reinterpret_cast<const float4*>(hI + fc[u]);
#pragma unroll
        for (int u = 0; u < QB; ++u) pi[u] = *reinterpret_cast<const float4*>(pI + fc[u]);
        if (DUAL) {
#pragma unroll
            for (int u = 0; u < QB; ++u) bi[u] = *reinterpret_cast<const float4*>(bI + fc[u]);
        }
        if (HASHD) {
#pragma unroll
            for (int u = 0; u < QB; ++u) ti[u] = *reinterpret_cast<const float4*>(tI + fc[u]);
        }
        __builtin_amdgcn_sched_barrier(0);                        // the batch's loads stay together, ahead of its products
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            const int q = q0 + u;
            const bool ok = va && 16 * q + 4 * lk + 4 <= F;
            wdb[4 * q] = ok ? pi[u].x * hj[u].x + pj[u].x * hi[u].x : 0.f;
            wdb[4 * q + 1] = ok ? pi[u].y * hj[u].y + pj[u].y * hi[u].y : 0.f;
            wdb[4 * q + 2] = ok ? pi[u].z * hj[u].z + pj[u].z * hi[u].z : 0.f;
            wdb[4 * q + 3] = ok ? pi[u].w * hj[u].w + pj[u].w * hi[u].w : 0.f;
            if (DUAL) {
                float4 w = {bi[u].x * hj[u].x + bj[u].x * hi[u].x, bi[u].y * hj[u].y + bj[u].y * hi[u].y,
                            bi[u].z * hj[u].z + bj[u].z * hi[u].z, bi[u].w * hj[u].w + bj[u].w * hi[u].w};
                if (HASHD) {
                    w.x += pi[u].x * tj[u].x + pj[u].x * ti[u].x; w.y += pi[u].y * tj[u].y + pj[u].y * ti[u].y;
                    w.z += pi[u].z * tj[u].z + pj[u].z * ti[u].z; w.w += pi[u].w * tj[u].w + pj[u].w * ti[u].w;
                }
                wb[4 * q] = ok ? w.x : 0.f; wb[4 * q + 1] = ok ? w.y : 0.f;
                wb[4 * q + 2] = ok ? w.z : 0.f; wb[4 * q + 3] = ok ? w.w : 0.f;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// physical row of filter f in the LDS copy of W2 used as the B operand of  [16 x F] x [F x G]:
// k-step ks = 4 (f / 16) + f % 4, k-lane lk = (f % 16) / 4
__host__ __device__ inline int w2_row(int f) { return (4 * (f >> 4) + (f & 3)) * 4 + ((f & 15) >> 2); }

template <int GP, int FT, bool DUAL, bool THETA>
// (f32 operands: the 64 adjoint-row registers stay live as MFMA operands and the weights take 71 KB of LDS, so the sweep with
//  parameter gradients keeps one wave per SIMD -- with four k-blocks in flight per round trip -- and the others two)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GP == 32 ? (THETA ? 1 : 2) : 1)))
void cfconv_bwd_kernel(const BwdArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int S1 = GP % 32 == 16 ? GP : GP + 16;
    constexpr int FP = 16 * FT;
    constexpr int SA = GP + 2;
    constexpr int SW = FP + 4;                                   // tile stride: 4 mod 32, rows 16-B aligned
    constexpr int KS = GP / 4, NT = GP / 16;
    constexpr int WSZ = THETA ? 16 * SW : 16 * SA;               // per-wave scratch (tile / transposes)
    static_assert(!THETA || DUAL, "parameter gradients come from the dual sweep");
    float* w1s = sm;                         // [GP][S1]  B[k][j] = W1[j][k]      (a = g W1^T)
    float* w1n = w1s + GP * S1;              // [GP][S1]  B[j][k] = W1[j][k]      (g_b = a_b W1)
    float* w2b = w1n + GP * S1;              // [FP][S1]  row w2_row(f): W2[f][k] (s_b = W_b W2)
    float* mus = w2b + FP * S1;
    float* cfs = mus + GP;
    float* c2s = cfs + GP;
    float* b1s = c2s + GP;
    float* wsc = b1s + GP;                   // [4 waves][WSZ]
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave index: uniform)
    const int G = A.net.G, F = A.net.F, RS = A.net.RS;
    for (int t = tid; t < GP * GP; t += 256) {
        const int k = t / GP, j = t % GP;
        const bool in = k < G && j < G;
        w1s[k * S1 + j] = in ? A.net.W1[j * G + k] : 0.f;
        w1n[k * S1 + j] = in ? A.net.W1[k * G + j] : 0.f;
    }
    for (int t = tid; t < FP * GP; t += 256) {
        const int f = t / GP, k = t % GP;
        w2b[w2_row(f) * S1 + k] = (f < F && k < G) ? A.net.W2[(size_t)f * G + k] : 0.f;
    }
    for (int k = tid; k < GP; k += 256) {
        const float c = k < G ? A.net.coef[k] : 0.f;
        mus[k] = k < G ? A.net.mu[k] : 0.f;
        cfs[k] = c * LOG2E;
        c2s[k] = 2.f * c;
        b1s[k] = k < G ? A.net.b1[k] : ((THETA && k == GP - 1) ? B2COL_BIAS : 0.f);     // (see B2COL_BIAS)
    }
    __syncthreads();

    const int li = lane & 15, lk = lane >> 4;
    float* ws = wsc + wid * WSZ;
    // persistent per-wave gradient accumulators (THETA)
    f32x4 gW2[THETA ? FT : 1][THETA ? NT : 1], gW1[THETA ? NT : 1][THETA ? NT : 1];
    float gb1[NT];
    float gmu[THETA ? NT : 1], gcf[THETA ? NT : 1];      // THETA: gradients of the Gaussian centres and coefficients (trainable smearing)
    if (THETA) {
#pragma unroll
        for (int c = 0; c < NT; ++c) gmu[c] = gcf[c] = 0.f;
#pragma unroll
        for (int a = 0; a < FT; ++a)
#pragma unroll
            for (int c = 0; c < NT; ++c) gW2[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int c = 0; c < NT; ++c) gW1[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int c = 0; c < NT; ++c) gb1[c] = 0.f;

    // 64 edges per workgroup step, 16 per wave; a fixed-capacity list is swept over its real rows only (the split over
    // the XCDs then stays balanced)
    const long long nrows = A.n_valid ? min((long long)*A.n_valid, A.E) : A.E;
    const long long ntiles = (nrows + 63) / 64;
    int t_begin, t_end, t_step;
    xcd_sweep((int)ntiles, 1, t_begin, t_end, t_step);           // the half list is sorted by atom: same locality argument
    const bool has_hd = A.hd != nullptr;                          // (wave-uniform: the first block's node rows have no tangent)
    EdgeIdx nx = load_edge_idx<DUAL>(A, (long long)t_begin * 64 + wid * 16 + li);
    for (long long tile = t_begin; tile < t_end; tile += t_step) {
        const long long e0 = tile * 64 + wid * 16;
        // ---- A-layout row: edge e0 + li (indices, distance and tangent were requested a tile ahead)
        const EdgeIdx cur = nx;
        if (tile + t_step < t_end) nx = load_edge_idx<DUAL>(A, (tile + t_step) * 64 + wid * 16 + li);
        const int ia = cur.i, ja = cur.j;
        const bool va = cur.d >= 0.f;                             // (-1: padding row, or a pair of a stored list beyond the cutoff now)
        if (__ballot(va) == 0ull) continue;                       // a wave's 16 rows all padding / past the end: nothing to add
        const float da = va ? cur.d : PAD_D;
        const float dda = (DUAL && va) ? cur.dd : 0.f;
        // ---- adjoint rows of the filter output as A fragments (k-step 4 q + c <-> filter 16 q + 4 lk + c)
        float wdb[4 * FT], wb[DUAL ? 4 * FT : 1];
        constexpr int QB = (GP == 64 && THETA) ? 2 : 4;   // k-blocks gathered per round trip (registers in flight)
        if (DUAL && has_hd) gather_adjoint_rows<FT, QB, DUAL, true>(A, ia, ja, va, lk, F, RS, wdb, wb);
        else gather_adjoint_rows<FT, QB, DUAL, false>(A, ia, ja, va, lk, F, RS, wdb, wb);
        // ---- recompute layer 1: a = g W1^T + b1 (and its tangent) in the accumulator layout
        float sg[NT][4], qd[DUAL ? NT : 1][4], sc[THETA ? NT : 1][4], sdc[THETA ? NT : 1][4];
        {
            float af[KS], adf[DUAL ? KS : 1];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int k = ks * 4 + lk;
                const float x = da - mus[k];
                const float g = __builtin_amdgcn_exp2f(cfs[k] * x * x);
                af[ks] = g;
                if (DUAL) adf[ks] = g * (c2s[k] * x) * dda;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const float bfr = w1s[(ks * 4 + lk) * S1 + nt * 16 + li];
                    acc = MFMA(af[ks], bfr, acc);
                    if (DUAL) accd = MFMA(adf[ks], bfr, accd);
                }
                const float bias = b1s[nt * 16 + li];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s, g1;
                    ssp_sig(acc[r] + bias, s, g1);
                    sg[nt][r] = g1;
                    if (DUAL) qd[nt][r] = g1 * (1.f - g1) * accd[r];
                    if (THETA) { sc[nt][r] = s; sdc[nt][r] = g1 * accd[r]; }
                }
            }
        }
        // ---- THETA: gW2[f][k] += sum_e Wdb[e][f] sd[e][k] + Wb[e][f] s[e][k]  (contraction over the 16 edges,
        //      edge = 4 lk + r: B operands are the accumulator-layout registers; A through the LDS tile)
        if (THETA) {
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int q = 0; q < FT; ++q)
                    *reinterpret_cast<float4*>(&ws[li * SW + 16 * q + 4 * lk]) =
                        pass == 0 ? make_float4(wdb[4 * q], wdb[4 * q + 1], wdb[4 * q + 2], wdb[4 * q + 3])
                                  : make_float4(wb[4 * q], wb[4 * q + 1], wb[4 * q + 2], wb[4 * q + 3]);
#pragma unroll
                for (int mt = 0; mt < FT; ++mt) {
                    float at[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) at[r] = ws[(4 * lk + r) * SW + mt * 16 + li];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            gW2[mt][nt] = MFMA(at[r], pass == 0 ? sdc[nt][r] : sc[nt][r], gW2[mt][nt]);
                }
            }
        }
        // ---- s_db = Wdb W2, s_b = Wb W2   ([16 x F] x [F x G])
        f32x4 sdb[NT], sb[DUAL ? NT : 1];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4 * FT; ++ks) {
                const float bfr = w2b[(ks * 4 + lk) * S1 + nt * 16 + li];
                acc = MFMA(wdb[ks], bfr, acc);
                if (DUAL) acc2 = MFMA(wb[ks], bfr, acc2);
            }
            sdb[nt] = acc;
            if (DUAL) sb[nt] = acc2;
        }
        // ---- through the shifted softplus: a_db = s_db sig(a) ; a_b = s_b sig(a) + s_db sig'(a) a_dot
        float adb[NT][4], ab[DUAL ? NT : 1][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                adb[nt][r] = sdb[nt][r] * sg[nt][r];
                if (DUAL) {
                    ab[nt][r] = sb[nt][r] * sg[nt][r] + sdb[nt][r] * qd[nt][r];
                    gb1[nt] += ab[nt][r];
                }
            }
        // ---- g_db = a_db W1, g_b = a_b W1: accumulator layout -> A layout through the wave's scratch
        f32x4 gdb[NT], gb[DUAL ? NT : 1];
#pragma unroll
        for (int pass = 0; pass < (DUAL ? 2 : 1); ++pass) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ws[(4 * lk + r) * SA + nt * 16 + li] = pass == 0 ? adb[nt][r] : ab[DUAL ? nt : 0][r];
            float af[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) af[ks] = ws[li * SA + ks * 4 + lk];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc = MFMA(af[ks], w1n[(ks * 4 + lk) * S1 + nt * 16 + li], acc);
                if (pass == 0) gdb[nt] = acc; else gb[DUAL ? nt : 0] = acc;
            }
        }
        // ---- contract with the Gaussian derivatives in the accumulator layout (rows 4 lk + r, Gaussian nt*16 + li)
        float s_dd[4] = {0.f, 0.f, 0.f, 0.f}, s_d[4] = {0.f, 0.f, 0.f, 0.f};
        float gc[THETA ? NT : 1][4], gdc[THETA ? NT : 1][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dr = __shfl(da, 4 * lk + r, 64);
            const float ddr = DUAL ? __shfl(dda, 4 * lk + r, 64) : 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int k = nt * 16 + li;
                const float x = dr - mus[k];
                const float g = __builtin_amdgcn_exp2f(cfs[k] * x * x);
                const float ph = c2s[k] * x;
                const float gp = g * ph;
                s_dd[r] = fmaf(gdb[nt][r], gp, s_dd[r]);
                if (DUAL) {
                    s_d[r] = fmaf(gb[nt][r], gp, s_d[r]);
                    s_d[r] = fmaf(gdb[nt][r] * ddr, g * (ph * ph + c2s[k]), s_d[r]);
                }
                if (THETA) {
                    gc[nt][r] = g; gdc[nt][r] = gp * ddr;
                    // g_k = exp(c_k x^2), x = d - mu_k:  dg/dmu = -dg/dd,  dg/dc = g x^2;  the tangent gd = g 2c x dd likewise
                    const float td_ = gdb[nt][r] * ddr;
                    gmu[nt] -= gb[nt][r] * gp + td_ * g * (ph * ph + c2s[k]);
                    gcf[nt] += gb[nt][r] * g * x * x + td_ * g * x * (x * ph + 2.f);
                }
            }
        }
        // ---- THETA: gW1[j][k] += sum_e a_db[e][j] gd[e][k] + a_b[e][j] g[e][k]  (all operands already in the
        //      accumulator layout: A = the adjoint of a, B = the Gaussians)
        if (THETA) {
#pragma unroll
            for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        gW1[mt][nt] = MFMA(adb[mt][r], gdc[nt][r], gW1[mt][nt]);
                        gW1[mt][nt] = MFMA(ab[mt][r], gc[nt][r], gW1[mt][nt]);
                    }
        }
        // ---- row sums over the 16 Gaussian lanes, then read-add-write of the edge's own entries
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                s_dd[r] += __shfl_xor(s_dd[r], o, 64);
                if (DUAL) s_d[r] += __shfl_xor(s_d[r], o, 64);
            }
        }
        if (li == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long e = e0 + 4 * lk + r;
                if (e < A.E) {
                    A.dd_b[e] += s_dd[r];
                    if (DUAL) A.d_b[e] += s_d[r];
                }
            }
        }
    }

    // ---- THETA: ordered cross-wave reduction through LDS, one partial record per workgroup
    if (THETA) {
        constexpr int REC0 = GP * GP + GP + FP * GP;             // [gW1 | gb1 | gW2 | gmu | gcoef], padded sizes
        constexpr int REC = REC0 + 2 * GP;
        __syncthreads();                                         // everyone is done with the weights / scratch
        float* buf = sm;                                         // REC floats: reuses the weight area (GP*S1*2 + FP*S1 >= REC)
#pragma unroll
        for (int v = 0; v < NT; ++v) {
            gb1[v] += __shfl_xor(gb1[v], 16, 64); gb1[v] += __shfl_xor(gb1[v], 32, 64);
            gmu[v] += __shfl_xor(gmu[v], 16, 64); gmu[v] += __shfl_xor(gmu[v], 32, 64);
            gcf[v] += __shfl_xor(gcf[v], 16, 64); gcf[v] += __shfl_xor(gcf[v], 32, 64);
        }
        for (int w = 0; w < 4; ++w) {
            if (wid == w) {
#pragma unroll
                for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int idx = (mt * 16 + 4 * lk + r) * GP + nt * 16 + li;       // gW1[j][k]
                            buf[idx] = (w ? buf[idx] : 0.f) + gW1[mt][nt][r];
                        }
                if (lk == 0) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int idx = GP * GP + nt * 16 + li;
                        buf[idx] = (w ? buf[idx] : 0.f) + gb1[nt];
                        const int im = REC0 + nt * 16 + li, ic = REC0 + GP + nt * 16 + li;
                        buf[im] = (w ? buf[im] : 0.f) + gmu[nt];
                        buf[ic] = (w ? buf[ic] : 0.f) + gcf[nt];
                    }
                }
#pragma unroll
                for (int mt = 0; mt < FT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int idx = GP * GP + GP + (mt * 16 + 4 * lk + r) * GP + nt * 16 + li;   // gW2[f][k]
                            buf[idx] = (w ? buf[idx] : 0.f) + gW2[mt][nt][r];
                        }
            }
            __syncthreads();
        }
        float* out = A.part + (size_t)blockIdx.x * REC;
        for (int t = tid; t < REC; t += 256) out[t] = buf[t];
    }
}

template <int GP, int FT>
size_t bwd_lds_bytes(bool theta) {
    constexpr int S1 = GP % 32 == 16 ? GP : GP + 16;
    constexpr int FP = 16 * FT;
    const size_t wsz = theta ? 16 * (FP + 4) : 16 * (GP + 2);
    return sizeof(float) * ((size_t)2 * GP * S1 + (size_t)FP * S1 + 4 * GP + 4 * wsz);
}

// ---------------------------------------------------------------------------------------------
// bf16-operand variant of the reverse sweep (BASELINE config #5: "bf16 cfconv MFMA, full fwd + adjoint"): the same
// sweep with every matrix product on bf16 operands and fp32 accumulation --
//   a = g W1^T, g_b = a_b W1          v_mfma_f32_16x16x32_bf16 (K = G: one or two instructions per 16 x 16 tile)
//   s_b = W_b W2                      v_mfma_f32_16x16x16_bf16: a lane's four consecutive filters 16 q + 4 lk + c ARE the
//                                     instruction's k = 4 lk + c, so the float4 gathers feed it without a permutation
//   gW2 += W_b^T s, gW1 += a_b^T g    v_mfma_f32_16x16x16_bf16 over the tile's 16 edges: the accumulator layout
//                                     (edge = 4 lk + r) is exactly its operand layout -- one instruction where the f32
//                                     kernel issues four
// 70 MFMAs per 16-edge tile instead of 352 (dual + theta, G <= 32, F = 128); W1 / W2 live in LDS as bf16 (48 KB per
// workgroup instead of 71: three resident workgroups per CU).  Gaussians, activations, the adjoint rows, every
// accumulation and the contraction with the Gaussian derivatives stay fp32; operands are rounded to nearest even.
typedef short bf16x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k((a), (b), (c), 0, 0, 0)
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d) {
    const u32x2v u = {cvt_pk_bf16(a, b), cvt_pk_bf16(c, d)};
    return __builtin_bit_cast(bf16x4, u);
}

// gather_adjoint_rows for the bf16 sweep: the products leave the gather as packed bf16 MFMA operands (2 registers per
// k-block and matrix instead of 4 + 4 floats kept for the whole tile) and, for the parameter gradients, go straight into
// the wave's two bf16 LDS tiles  tdb / tb [16 edges][SWB]  that the transposed operand of gW2 is read from.
template <int FT, int QB, int SWB, bool DUAL, bool HASHD, bool TILE>
__device__ __forceinline__ void gather_adjoint_rows_bf16(const BwdArgs& A, int ia, int ja, bool va, int li, int lk, int F,
                                                         int RS, bf16x4 (&wdbp)[FT], bf16x4 (&wbp)[DUAL ? FT : 1],
                                                         unsigned short* tdb, unsigned short* tb) {
    const long long ri = (long long)(va ? ia : 0) * RS + 4 * lk, rj = (long long)(va ? ja : 0) * RS + 4 * lk;
    const float* __restrict__ hI = A.h + ri;
    const float* __restrict__ hJ = A.h + rj;
    const float* __restrict__ pI = A.mdb + ri;
    const float* __restrict__ pJ = A.mdb + rj;
    const float* __restrict__ bI = DUAL ? A.mb + ri : nullptr;
    const float* __restrict__ bJ = DUAL ? A.mb + rj : nullptr;
    const float* __restrict__ tI = HASHD ? A.hd + ri : nullptr;
    const float* __restrict__ tJ = HASHD ? A.hd + rj : nullptr;
#pragma unroll
    for (int q0 = 0; q0 < FT; q0 += QB) {
        float4 hi[QB], hj[QB], pi[QB], pj[QB], bi[DUAL ? QB : 1], bj[DUAL ? QB : 1], ti[HASHD ? QB : 1], tj[HASHD ? QB : 1];
        // (array by array: the 64-byte pieces two neighbouring k-blocks take of a row are the halves of one 128-byte line --
        //  requested back to back, the second is served by the fill the first started)
        int fc[QB];
#pragma unroll
        for (int u = 0; u < QB; ++u) fc[u] = 16 * (q0 + u) + 4 * lk + 4 <= F ? 16 * (q0 + u) : 0;
#pragma unroll
        for (int u = 0; u < QB; ++u) hj[u] = *reinterpret_cast<const float4*>(hJ + fc[u]);
#pragma unroll
        for (int u = 0; u < QB; ++u) pj[u] = *reinterpret_cast<const float4*>(pJ + fc[u]);
        if (DUAL) {
#pragma unroll
            for (int u = 0; u < QB; ++u) bj[u] = *reinterpret_cast<const float4*>(bJ + fc[u]);
        }
        if (HASHD) {
#pragma unroll
            for (int u = 0; u < QB; ++u) tj[u] = *reinterpret_cast<const float4*>(tJ + fc[u]);
        }
#pragma unroll
        for (int u = 0; u < QB; ++u) hi[u] = *reinterpret_cast<const float4*>(hI + fc[u]);
#pragma unroll
        for (int u = 0; u < QB; ++u) pi[u] = *reinterpret_cast<const float4*>(pI + fc[u]);
        if (DUAL) {
#pragma unroll
            for (int u = 0; u < QB; ++u) bi[u] = *reinterpret_cast<const float4*>(bI + fc[u]);
        }
        if (HASHD) {
#pragma unroll
            for (int u = 0; u < QB; ++u) ti[u] = *reinterpret_cast<const float4*>(tI + fc[u]);
        }
        __builtin_amdgcn_sched_barrier(0);                        // the batch's loads stay together, ahead of its products
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            const int q = q0 + u;
            const bool ok = va && 16 * q + 4 * lk + 4 <= F;
            const bf16x4 zero4 = {0, 0, 0, 0};
            const bf16x4 a = pack4(pi[u].x * hj[u].x + pj[u].x * hi[u].x, pi[u].y * hj[u].y + pj[u].y * hi[u].y,
                                   pi[u].z * hj[u].z + pj[u].z * hi[u].z, pi[u].w * hj[u].w + pj[u].w * hi[u].w);
            wdbp[q] = ok ? a : zero4;
            if (TILE) *reinterpret_cast<bf16x4*>(&tdb[li * SWB + 16 * q + 4 * lk]) = wdbp[q];
            if (DUAL) {
                float4 w = {bi[u].x * hj[u].x + bj[u].x * hi[u].x, bi[u].y * hj[u].y + bj[u].y * hi[u].y,
                            bi[u].z * hj[u].z + bj[u].z * hi[u].z, bi[u].w * hj[u].w + bj[u].w * hi[u].w};
                if (HASHD) {
                    w.x += pi[u].x * tj[u].x + pj[u].x * ti[u].x; w.y += pi[u].y * tj[u].y + pj[u].y * ti[u].y;
                    w.z += pi[u].z * tj[u].z + pj[u].z * ti[u].z; w.w += pi[u].w * tj[u].w + pj[u].w * ti[u].w;
                }
                const bf16x4 b = pack4(w.x, w.y, w.z, w.w);
                wbp[q] = ok ? b : zero4;
                if (TILE) *reinterpret_cast<bf16x4*>(&tb[li * SWB + 16 * q + 4 * lk]) = wbp[q];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Filter owned by (k-block q, k-lane lk, component c) is  fbase + c.  With f32 rows a lane's 16-byte load is ONE k-block's four
// filters (16 q + 4 lk); with bf16 rows (R16) a 16-byte load carries EIGHT filters, the pieces of k-blocks 2 Q and 2 Q + 1, so
// the contraction index is permuted to  32 Q + 8 lk + 4 (q & 1) + c  -- on the adjoint rows, on the W2 operand read from LDS
// and on the columns of the gW2 tiles alike (any permutation of a contraction index is as good as another).
template <bool R16>
__device__ __forceinline__ int fbase(int q, int lk) { return R16 ? 32 * (q >> 1) + 8 * lk + 4 * (q & 1) : 16 * q + 4 * lk; }

__device__ __forceinline__ float4 widen4(const uint4& x, int half) {       // filters 4 half .. 4 half + 3 of a lane's eight
    const unsigned a = half ? x.z : x.x, b = half ? x.w : x.y;
    return make_float4(__uint_as_float(a << 16), __uint_as_float(a & 0xffff0000u), __uint_as_float(b << 16),
                       __uint_as_float(b & 0xffff0000u));
}

// gather_adjoint_rows_bf16 over bf16 MIRRORS of the node matrices (mdg_cfconv_bwd_rows16): half the loads, half the lines.
// QB k-blocks per batch = QB / 2 loads per matrix; the widened values enter the same f32 products.
template <int FT, int QB, int SWB, bool DUAL, bool HASHD, bool TILE>
__device__ __forceinline__ void gather_adjoint_rows_r16(const BwdArgs& A, int ia, int ja, bool va, int li, int lk, int F,
                                                        int RS16, bf16x4 (&wdbp)[FT], bf16x4 (&wbp)[DUAL ? FT : 1],
                                                        unsigned short* tdb, unsigned short* tb) {
    static_assert(QB % 2 == 0 && FT % QB == 0, "k-blocks come in pairs");
    constexpr int NL = QB / 2;
    const unsigned ri = (unsigned)(va ? ia : 0) * (unsigned)RS16 + 8u * lk, rj = (unsigned)(va ? ja : 0) * (unsigned)RS16 + 8u * lk;
    const unsigned short* __restrict__ hI = reinterpret_cast<const unsigned short*>(A.h) + ri;
    const unsigned short* __restrict__ hJ = reinterpret_cast<const unsigned short*>(A.h) + rj;
    const unsigned short* __restrict__ pI = reinterpret_cast<const unsigned short*>(A.mdb) + ri;
    const unsigned short* __restrict__ pJ = reinterpret_cast<const unsigned short*>(A.mdb) + rj;
    const unsigned short* __restrict__ bI = DUAL ? reinterpret_cast<const unsigned short*>(A.mb) + ri : nullptr;
    const unsigned short* __restrict__ bJ = DUAL ? reinterpret_cast<const unsigned short*>(A.mb) + rj : nullptr;
    const unsigned short* __restrict__ tI = HASHD ? reinterpret_cast<const unsigned short*>(A.hd) + ri : nullptr;
    const unsigned short* __restrict__ tJ = HASHD ? reinterpret_cast<const unsigned short*>(A.hd) + rj : nullptr;
#pragma unroll
    for (int q0 = 0; q0 < FT; q0 += QB) {
        uint4 hi[NL], hj[NL], pi[NL], pj[NL], bi[DUAL ? NL : 1], bj[DUAL ? NL : 1], ti[HASHD ? NL : 1], tj[HASHD ? NL : 1];
        int fc[NL];
#pragma unroll
        for (int u = 0; u < NL; ++u) fc[u] = 32 * (q0 / 2 + u) + 8 * lk + 8 <= F ? 32 * (q0 / 2 + u) : 0;
#pragma unroll
        for (int u = 0; u < NL; ++u) hj[u] = *reinterpret_cast<const uint4*>(hJ + fc[u]);
#pragma unroll
        for (int u = 0; u < NL; ++u) pj[u] = *reinterpret_cast<const uint4*>(pJ + fc[u]);
        if (DUAL) {
#pragma unroll
            for (int u = 0; u < NL; ++u) bj[u] = *reinterpret_cast<const uint4*>(bJ + fc[u]);
        }
        if (HASHD) {
#pragma unroll
            for (int u = 0; u < NL; ++u) tj[u] = *reinterpret_cast<const uint4*>(tJ + fc[u]);
        }
#pragma unroll
        for (int u = 0; u < NL; ++u) hi[u] = *reinterpret_cast<const uint4*>(hI + fc[u]);
#pragma unroll
        for (int u = 0; u < NL; ++u) pi[u] = *reinterpret_cast<const uint4*>(pI + fc[u]);
        if (DUAL) {
#pragma unroll
            for (int u = 0; u < NL; ++u) bi[u] = *reinterpret_cast<const uint4*>(bI + fc[u]);
        }
        if (HASHD) {
#pragma unroll
            for (int u = 0; u < NL; ++u) ti[u] = *reinterpret_cast<const uint4*>(tI + fc[u]);
        }
        __builtin_amdgcn_sched_barrier(0);                        // the batch's loads stay together, ahead of its products
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            const int q = q0 + u, fb = fbase<true>(q, lk);
            const bool ok = va && fb + 4 <= F;
            const bf16x4 zero4 = {0, 0, 0, 0};
            const float4 Hi = widen4(hi[u >> 1], u & 1), Hj = widen4(hj[u >> 1], u & 1);
            const float4 Pi = widen4(pi[u >> 1], u & 1), Pj = widen4(pj[u >> 1], u & 1);
            const bf16x4 a = pack4(Pi.x * Hj.x + Pj.x * Hi.x, Pi.y * Hj.y + Pj.y * Hi.y, Pi.z * Hj.z + Pj.z * Hi.z,
                                   Pi.w * Hj.w + Pj.w * Hi.w);
            wdbp[q] = ok ? a : zero4;
            if (TILE) *reinterpret_cast<bf16x4*>(&tdb[li * SWB + fb]) = wdbp[q];
            if (DUAL) {
                const float4 Bi = widen4(bi[u >> 1], u & 1), Bj = widen4(bj[u >> 1], u & 1);
                float4 w = {Bi.x * Hj.x + Bj.x * Hi.x, Bi.y * Hj.y + Bj.y * Hi.y, Bi.z * Hj.z + Bj.z * Hi.z, Bi.w * Hj.w + Bj.w * Hi.w};
                if (HASHD) {
                    const float4 Ti = widen4(ti[u >> 1], u & 1), Tj = widen4(tj[u >> 1], u & 1);
                    w.x += Pi.x * Tj.x + Pj.x * Ti.x; w.y += Pi.y * Tj.y + Pj.y * Ti.y;
                    w.z += Pi.z * Tj.z + Pj.z * Ti.z; w.w += Pi.w * Tj.w + Pj.w * Ti.w;
                }
                const bf16x4 b = pack4(w.x, w.y, w.z, w.w);
                wbp[q] = ok ? b : zero4;
                if (TILE) *reinterpret_cast<bf16x4*>(&tb[li * SWB + fb]) = wbp[q];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int GP, int FT, bool DUAL, bool THETA, bool R16 = false>
// (waves per SIMD: round 3 measured more waves as a loss -- with each k-block's gathers behind a divergent branch the sweep
//  made 8-16 dependent round trips per tile whatever the occupancy, and the registers a second wave needed were spilled.  With
//  the batched unconditional gathers of gather_adjoint_rows_bf16 the state fits two waves: LDS allows three workgroups per CU)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GP == 32 ? (THETA ? 2 : (DUAL ? (R16 ? 3 : 2) : (R16 ? 4 : 3))) : 1)))
void cfconv_bwd_bf16_kernel(const BwdArgs A) {
    static_assert(!R16 || FT == 8, "bf16 node rows: layers of more than 64 filters");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int FP = 16 * FT;
    constexpr int KSB = GP + 8;                                  // bf16 row stride of the W1 copies (16-B aligned rows)
    constexpr int FS = FP + 8;                                   // bf16 row stride of the W2 copy
    constexpr int SA = GP + 4;                                   // f32 scratch stride (rows 16-B aligned)
    constexpr int SW = FP + 4;                                   // f32 words of the per-wave scratch per tile row (THETA)
    constexpr int SWB = FP + 4;                                  // bf16 tile stride (elements): rows 8-B aligned, 4 rows = 32 B mod 128
    constexpr int KB = GP / 32, NT = GP / 16;
    constexpr int WSZ = THETA ? 16 * SW : 16 * SA;               // per-wave scratch (two bf16 tiles / transposes), floats
    static_assert(!THETA || DUAL, "parameter gradients come from the dual sweep");
    static_assert(!THETA || SW >= SA, "the tile scratch also serves the transposes");
    static_assert(2 * SWB * 2 <= SW * 4, "two bf16 tiles fit the scratch of one f32 tile");
    float* mus = sm;
    float* cfs = mus + GP;
    float* c2s = cfs + GP;
    float* b1s = c2s + GP;
    float* wsc = b1s + GP;                                       // [4 waves][WSZ]
    unsigned short* w1b = reinterpret_cast<unsigned short*>(wsc + 4 * WSZ);   // [GP j][KSB]  W1[j][k]: B of a = g W1^T
    unsigned short* w1nb = w1b + GP * KSB;                       // [GP n][KSB]  W1[j][n]: B of g_b = a_b W1  (k = j)
    unsigned short* w2g = w1nb + GP * KSB;                       // [GP k][FS]   W2[f][k]: B of s_b = W_b W2  (k = f)
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave index: uniform)
    const int G = A.net.G, F = A.net.F, RS = A.net.RS;
    for (int t = tid; t < GP * GP; t += 256) {
        const int j = t / GP, c = t % GP;
        w1b[j * KSB + c] = (j < G && c < G) ? f2bf(A.net.W1[j * G + c]) : 0;       // row j, column k = c
        w1nb[j * KSB + c] = (j < G && c < G) ? f2bf(A.net.W1[c * G + j]) : 0;      // row n = j, column j' = c: W1[j'][n]
    }
    for (int t = tid; t < FP * GP; t += 256) {
        const int f = t / GP, c = t % GP;                        // (consecutive threads: consecutive k of one W2 row)
        w2g[c * FS + f] = (f < F && c < G) ? f2bf(A.net.W2[(size_t)f * G + c]) : 0;
    }
    for (int c = tid; c < GP; c += 256) {
        const float cc = c < G ? A.net.coef[c] : 0.f;
        mus[c] = c < G ? A.net.mu[c] : 0.f;
        cfs[c] = cc * LOG2E;
        c2s[c] = 2.f * cc;
        b1s[c] = c < G ? A.net.b1[c] : ((THETA && c == GP - 1) ? B2COL_BIAS : 0.f);     // (see B2COL_BIAS)
    }
    __syncthreads();

    const int li = lane & 15, lk = lane >> 4;
    float* ws = wsc + wid * WSZ;
    // persistent per-wave gradient accumulators (THETA)
    f32x4 gW2[THETA ? FT : 1][THETA ? NT : 1], gW1[THETA ? NT : 1][THETA ? NT : 1];
    float gb1[NT];
    float gmu[THETA ? NT : 1], gcf[THETA ? NT : 1];      // THETA: gradients of the Gaussian centres and coefficients (trainable smearing)
    if (THETA) {
#pragma unroll
        for (int c = 0; c < NT; ++c) gmu[c] = gcf[c] = 0.f;
#pragma unroll
        for (int a = 0; a < FT; ++a)
#pragma unroll
            for (int c = 0; c < NT; ++c) gW2[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int c = 0; c < NT; ++c) gW1[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int c = 0; c < NT; ++c) gb1[c] = 0.f;

    // 64 edges per workgroup step, 16 per wave; a fixed-capacity list is swept over its real rows only (the split over
    // the XCDs then stays balanced)
    const long long nrows = A.n_valid ? min((long long)*A.n_valid, A.E) : A.E;
    const long long ntiles = (nrows + 63) / 64;
    int t_begin, t_end, t_step;
    xcd_sweep((int)ntiles, 1, t_begin, t_end, t_step);           // the half list is sorted by atom: same locality argument
    const bool has_hd = A.hd != nullptr;                          // (wave-uniform: the first block's node rows have no tangent)
    EdgeIdx nx = load_edge_idx<DUAL>(A, (long long)t_begin * 64 + wid * 16 + li);
    for (long long tile = t_begin; tile < t_end; tile += t_step) {
        const long long e0 = tile * 64 + wid * 16;
        // ---- A-layout row: edge e0 + li (indices, distance and tangent were requested a tile ahead)
        const EdgeIdx cur = nx;
        if (tile + t_step < t_end) nx = load_edge_idx<DUAL>(A, (tile + t_step) * 64 + wid * 16 + li);
        const int ia = cur.i, ja = cur.j;
        const bool va = cur.d >= 0.f;                             // (-1: padding row, or a pair of a stored list beyond the cutoff now)
        if (__ballot(va) == 0ull) continue;                       // a wave's 16 rows all padding / past the end: nothing to add
        const float da = va ? cur.d : PAD_D;
        const float dda = (DUAL && va) ? cur.dd : 0.f;
        // ---- adjoint rows of the filter output as A fragments (k-step 4 q + c <-> filter 16 q + 4 lk + c)
        bf16x4 wdbp[FT], wbp[DUAL ? FT : 1];
        unsigned short* tdb = reinterpret_cast<unsigned short*>(ws);      // THETA: [16 edges][SWB] Wdb and, behind it, Wb
        unsigned short* tb = tdb + 16 * SWB;
        constexpr int QB = DUAL ? (THETA ? 2 : 4) : FT;          // k-blocks gathered per round trip (registers in flight)
        if constexpr (R16) {
            constexpr int QR = DUAL ? 4 : FT;                    // (a 16-byte load carries two k-blocks; the dual sweep without
                                                                 //  parameter gradients then fits three waves per SIMD: 196 -> 156 registers)
            if (DUAL && has_hd) gather_adjoint_rows_r16<FT, QR, SWB, DUAL, true, THETA>(A, ia, ja, va, li, lk, F, A.net.RS16, wdbp, wbp, tdb, tb);
            else gather_adjoint_rows_r16<FT, QR, SWB, DUAL, false, THETA>(A, ia, ja, va, li, lk, F, A.net.RS16, wdbp, wbp, tdb, tb);
        } else {
        if (DUAL && has_hd) gather_adjoint_rows_bf16<FT, QB, SWB, DUAL, true, THETA>(A, ia, ja, va, li, lk, F, RS, wdbp, wbp, tdb, tb);
        else gather_adjoint_rows_bf16<FT, QB, SWB, DUAL, false, THETA>(A, ia, ja, va, li, lk, F, RS, wdbp, wbp, tdb, tb);
        }
        // ---- recompute layer 1: a = g W1^T + b1 (and its tangent) in the accumulator layout
        float sg[NT][4], qd[DUAL ? NT : 1][4], sc[THETA ? NT : 1][4], sdc[THETA ? NT : 1][4];
        {
            bf16x8 af[KB], adf[DUAL ? KB : 1];
#pragma unroll
            for (int ks = 0; ks < KB; ++ks) {
                float gk[8], gdk[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int kk = ks * 32 + lk * 8 + t;
                    const float x = da - mus[kk];
                    const float g = __builtin_amdgcn_exp2f(cfs[kk] * x * x);
                    gk[t] = g;
                    gdk[t] = DUAL ? g * (c2s[kk] * x) * dda : 0.f;
                }
                af[ks] = pack8(gk);
                if (DUAL) adf[ks] = pack8(gdk);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KB; ++ks) {
                    const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(&w1b[(nt * 16 + li) * KSB + ks * 32 + lk * 8]);
                    acc = MFMA32(af[ks], bfr, acc);
                    if (DUAL) accd = MFMA32(adf[ks], bfr, accd);
                }
                const float bias = b1s[nt * 16 + li];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s, g1;
                    ssp_sig(acc[r] + bias, s, g1);
                    sg[nt][r] = g1;
                    if (DUAL) qd[nt][r] = g1 * (1.f - g1) * accd[r];
                    if (THETA) { sc[nt][r] = s; sdc[nt][r] = g1 * accd[r]; }
                }
            }
        }
        // ---- THETA: gW2[f][k] += sum_e Wdb[e][f] sd[e][k] + Wb[e][f] s[e][k]  (contraction over the 16 edges,
        //      edge = 4 lk + r: B operands are the accumulator-layout registers; A through the LDS tile)
        if (THETA) {
            bf16x4 sdp[NT], scp[NT];                                 // B[e = 4 lk + c][k]: the accumulator-layout registers, packed
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                sdp[nt] = pack4(sdc[nt][0], sdc[nt][1], sdc[nt][2], sdc[nt][3]);
                scp[nt] = pack4(sc[nt][0], sc[nt][1], sc[nt][2], sc[nt][3]);
            }
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const unsigned short* tl = pass == 0 ? tdb : tb;            // (written by the gather; same wave: program order)
#pragma unroll
                for (int mt = 0; mt < FT; ++mt) {
                    unsigned int at[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) at[r] = tl[(4 * lk + r) * SWB + mt * 16 + li];
                    const u32x2v apu = {at[0] | (at[1] << 16), at[2] | (at[3] << 16)};
                    const bf16x4 ap = __builtin_bit_cast(bf16x4, apu);      // A[f][e = 4 lk + c]
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        gW2[mt][nt] = MFMA16(ap, pass == 0 ? sdp[nt] : scp[nt], gW2[mt][nt]);
                }
            }
        }
        // ---- s_db = Wdb W2, s_b = Wb W2   ([16 x F] x [F x G])
        f32x4 sdb[NT], sb[DUAL ? NT : 1];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < FT; ++q) {                         // k = filter fbase(q, lk) + c on both operands
                const bf16x4 bfr = *reinterpret_cast<const bf16x4*>(&w2g[(nt * 16 + li) * FS + fbase<R16>(q, lk)]);
                acc = MFMA16(wdbp[q], bfr, acc);
                if (DUAL) acc2 = MFMA16(wbp[q], bfr, acc2);
            }
            sdb[nt] = acc;
            if (DUAL) sb[nt] = acc2;
        }
        // ---- through the shifted softplus: a_db = s_db sig(a) ; a_b = s_b sig(a) + s_db sig'(a) a_dot
        float adb[NT][4], ab[DUAL ? NT : 1][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                adb[nt][r] = sdb[nt][r] * sg[nt][r];
                if (DUAL) {
                    ab[nt][r] = sb[nt][r] * sg[nt][r] + sdb[nt][r] * qd[nt][r];
                    gb1[nt] += ab[nt][r];
                }
            }
        // ---- g_db = a_db W1, g_b = a_b W1: accumulator layout -> A layout through the wave's scratch
        f32x4 gdb[NT], gb[DUAL ? NT : 1];
#pragma unroll
        for (int pass = 0; pass < (DUAL ? 2 : 1); ++pass) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ws[(4 * lk + r) * SA + nt * 16 + li] = pass == 0 ? adb[nt][r] : ab[DUAL ? nt : 0][r];
            bf16x8 af[KB];
#pragma unroll
            for (int ks = 0; ks < KB; ++ks) {
                const float4 u0 = *reinterpret_cast<const float4*>(&ws[li * SA + ks * 32 + lk * 8]);
                const float4 u1 = *reinterpret_cast<const float4*>(&ws[li * SA + ks * 32 + lk * 8 + 4]);
                const float uu[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
                af[ks] = pack8(uu);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KB; ++ks)
                    acc = MFMA32(af[ks], *reinterpret_cast<const bf16x8*>(&w1nb[(nt * 16 + li) * KSB + ks * 32 + lk * 8]), acc);
                if (pass == 0) gdb[nt] = acc; else gb[DUAL ? nt : 0] = acc;
            }
        }
        // ---- contract with the Gaussian derivatives in the accumulator layout (rows 4 lk + r, Gaussian nt*16 + li)
        float s_dd[4] = {0.f, 0.f, 0.f, 0.f}, s_d[4] = {0.f, 0.f, 0.f, 0.f};
        float gc[THETA ? NT : 1][4], gdc[THETA ? NT : 1][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dr = __shfl(da, 4 * lk + r, 64);
            const float ddr = DUAL ? __shfl(dda, 4 * lk + r, 64) : 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int k = nt * 16 + li;
                const float x = dr - mus[k];
                const float g = __builtin_amdgcn_exp2f(cfs[k] * x * x);
                const float ph = c2s[k] * x;
                const float gp = g * ph;
                s_dd[r] = fmaf(gdb[nt][r], gp, s_dd[r]);
                if (DUAL) {
                    s_d[r] = fmaf(gb[nt][r], gp, s_d[r]);
                    s_d[r] = fmaf(gdb[nt][r] * ddr, g * (ph * ph + c2s[k]), s_d[r]);
                }
                if (THETA) {
                    gc[nt][r] = g; gdc[nt][r] = gp * ddr;
                    // g_k = exp(c_k x^2), x = d - mu_k:  dg/dmu = -dg/dd,  dg/dc = g x^2;  the tangent gd = g 2c x dd likewise
                    const float td_ = gdb[nt][r] * ddr;
                    gmu[nt] -= gb[nt][r] * gp + td_ * g * (ph * ph + c2s[k]);
                    gcf[nt] += gb[nt][r] * g * x * x + td_ * g * x * (x * ph + 2.f);
                }
            }
        }
        // ---- THETA: gW1[j][k] += sum_e a_db[e][j] gd[e][k] + a_b[e][j] g[e][k]  (all operands already in the
        //      accumulator layout: A = the adjoint of a, B = the Gaussians)
        if (THETA) {
#pragma unroll
            for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                {
                    gW1[mt][nt] = MFMA16(pack4(adb[mt][0], adb[mt][1], adb[mt][2], adb[mt][3]),
                                         pack4(gdc[nt][0], gdc[nt][1], gdc[nt][2], gdc[nt][3]), gW1[mt][nt]);
                    gW1[mt][nt] = MFMA16(pack4(ab[mt][0], ab[mt][1], ab[mt][2], ab[mt][3]),
                                         pack4(gc[nt][0], gc[nt][1], gc[nt][2], gc[nt][3]), gW1[mt][nt]);
                }
        }
        // ---- row sums over the 16 Gaussian lanes, then read-add-write of the edge's own entries
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                s_dd[r] += __shfl_xor(s_dd[r], o, 64);
                if (DUAL) s_d[r] += __shfl_xor(s_d[r], o, 64);
            }
        }
        if (li == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long e = e0 + 4 * lk + r;
                if (e < A.E) {
                    A.dd_b[e] += s_dd[r];
                    if (DUAL) A.d_b[e] += s_d[r];
                }
            }
        }
    }

    // ---- THETA: ordered cross-wave reduction through LDS, one partial record per workgroup
    if (THETA) {
        constexpr int REC0 = GP * GP + GP + FP * GP;             // [gW1 | gb1 | gW2 | gmu | gcoef], padded sizes
        constexpr int REC = REC0 + 2 * GP;
        __syncthreads();                                         // everyone is done with the weights / scratch
        float* buf = sm;                                         // REC floats: reuses the whole LDS block (bwd_bf16_lds_bytes reserves >= REC)
#pragma unroll
        for (int v = 0; v < NT; ++v) {
            gb1[v] += __shfl_xor(gb1[v], 16, 64); gb1[v] += __shfl_xor(gb1[v], 32, 64);
            gmu[v] += __shfl_xor(gmu[v], 16, 64); gmu[v] += __shfl_xor(gmu[v], 32, 64);
            gcf[v] += __shfl_xor(gcf[v], 16, 64); gcf[v] += __shfl_xor(gcf[v], 32, 64);
        }
        for (int w = 0; w < 4; ++w) {
            if (wid == w) {
#pragma unroll
                for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int idx = (mt * 16 + 4 * lk + r) * GP + nt * 16 + li;       // gW1[j][k]
                            buf[idx] = (w ? buf[idx] : 0.f) + gW1[mt][nt][r];
                        }
                if (lk == 0) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int idx = GP * GP + nt * 16 + li;
                        buf[idx] = (w ? buf[idx] : 0.f) + gb1[nt];
                        const int im = REC0 + nt * 16 + li, ic = REC0 + GP + nt * 16 + li;
                        buf[im] = (w ? buf[im] : 0.f) + gmu[nt];
                        buf[ic] = (w ? buf[ic] : 0.f) + gcf[nt];
                    }
                }
#pragma unroll
                for (int mt = 0; mt < FT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int idx = GP * GP + GP + (mt * 16 + 4 * lk + r) * GP + nt * 16 + li;   // gW2[f][k]
                            buf[idx] = (w ? buf[idx] : 0.f) + gW2[mt][nt][r];
                        }
            }
            __syncthreads();
        }
        float* out = A.part + (size_t)blockIdx.x * REC;
        for (int t = tid; t < REC; t += 256) out[t] = buf[t];
    }
}

template <int GP, int FT>
size_t bwd_bf16_lds_bytes(bool theta) {
    constexpr int FP = 16 * FT, KSB = GP + 8, FS = FP + 8, SA = GP + 4, SW = FP + 4;
    const size_t wsz = theta ? 16 * SW : 16 * SA;
    size_t b = sizeof(float) * (4 * GP + 4 * wsz) + sizeof(unsigned short) * ((size_t)2 * GP * KSB + (size_t)GP * FS);
    const size_t rec = sizeof(float) * ((size_t)GP * GP + 3 * GP + (size_t)FP * GP);
    if (theta && b < rec) b = rec;
    return (b + 15) / 16 * 16;
}

// sum of the per-workgroup partial records in a fixed order; un-pads [GP x GP | GP | FP x GP] to the true sizes.
// A workgroup owns 64 consecutive entries of the record (coalesced rows of `part`); its 16 waves take the records
// p = wave, wave + 16, ... with the loads of a wave independent of one another, then add up across waves in wave order.
constexpr int RED_WAVES = 16;
__global__ __launch_bounds__(64 * RED_WAVES) void cfconv_bwd_reduce_kernel(
    const float* __restrict__ part, int nrec, int GP, int FP, int G, int F, float* __restrict__ gW1, float* __restrict__ gb1,
    float* __restrict__ gW2, float* __restrict__ gmu, float* __restrict__ gcoef, int accumulate, float* __restrict__ gb2) {
    __shared__ float red[RED_WAVES][64];
    const int REC0 = GP * GP + GP + FP * GP, REC = REC0 + 2 * GP;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (t < REC) {
        int p = wid;
        for (; p + 3 * RED_WAVES < nrec; p += 4 * RED_WAVES) {
            s0 += part[(size_t)p * REC + t];
            s1 += part[(size_t)(p + RED_WAVES) * REC + t];
            s2 += part[(size_t)(p + 2 * RED_WAVES) * REC + t];
            s3 += part[(size_t)(p + 3 * RED_WAVES) * REC + t];
        }
        for (; p < nrec; p += RED_WAVES) s0 += part[(size_t)p * REC + t];
    }
    red[wid][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wid || t >= REC) return;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < RED_WAVES; ++w) s += red[w][lane];
    if (t < GP * GP) {
        const int j = t / GP, k = t % GP;
        if (j < G && k < G) gW1[j * G + k] = accumulate ? gW1[j * G + k] + s : s;     // (later filter chunks add)
    } else if (t < GP * GP + GP) {
        const int j = t - GP * GP;
        if (j < G) gb1[j] = accumulate ? gb1[j] + s : s;
    } else if (t < REC0) {
        const int u = t - GP * GP - GP, f = u / GP, k = u % GP;
        if (f < F && k < G) gW2[(size_t)f * G + k] = s;
        if (gb2 && f < F && k == GP - 1 && G < GP) gb2[f] = s;      // the bias column (B2COL_BIAS)
    } else {
        const int k = (t - REC0) % GP;
        float* out = t - REC0 < GP ? gmu : gcoef;                 // (nullable: asked for only with trainable smearing)
        if (out && k < G) out[k] = accumulate ? out[k] + s : s;
    }
}

// ============================================================================================ geometry
// d_e = |x_i - x_j - o_e| (nff/nn/models/schnet.py:142, raw image flags by default), unit vector, and for the
// tangent sweep dd_e = uhat . (w_i - w_j).  Padding rows (i = -1): |delta| = |o| (= 1e4), tangent 0.
// MASK: the list was searched with a skin (mdg_nbr_verlet_rebuild); a pair counts only while the builders' own test holds at
// the current positions -- D = x_j - x_i, reference minimum image, un-contracted d^2 < rc^2 and != 0 (topology.py:59-67):
// the same arithmetic, so the pair set is the one a fresh search at the cutoff finds.  Pairs outside get d = -1.
// (the geometry of one edge; products and sums are kept un-contracted so that the bits do not depend on the surrounding code)
struct EdgeGeo { float d, dd, ux, uy, uz, wx, wy, wz; };
template <bool MASK, bool DIAG>
__device__ __forceinline__ EdgeGeo edge_geometry(const float* __restrict__ x, const float* __restrict__ w,
                                                 const int64_t* __restrict__ nbr, const float* __restrict__ off, long long e,
                                                 const MdgCell& cell, float rc2) {
#pragma clang fp contract(off)
    const long long i = nbr[2 * e], j = nbr[2 * e + 1];
    bool masked = false;
    if (MASK && i >= 0) {
        float bx = x[3 * j] - x[3 * i], by = x[3 * j + 1] - x[3 * i + 1], bz = x[3 * j + 2] - x[3 * i + 2];
        min_image<DIAG>(cell, bx, by, bz);
        const float b2 = norm2_ref(bx, by, bz);
        masked = !((b2 < rc2) && (b2 != 0.f));
    }
    float dx = -off[3 * e], dy = -off[3 * e + 1], dz = -off[3 * e + 2];
    EdgeGeo g{};
    if (i >= 0) {
        dx += x[3 * i] - x[3 * j]; dy += x[3 * i + 1] - x[3 * j + 1]; dz += x[3 * i + 2] - x[3 * j + 2];
        if (w) { g.wx = w[3 * i] - w[3 * j]; g.wy = w[3 * i + 1] - w[3 * j + 1]; g.wz = w[3 * i + 2] - w[3 * j + 2]; }
    }
    const float r = sqrtf(dx * dx + dy * dy + dz * dz);
    const float ir = 1.0f / r;
    g.ux = dx * ir; g.uy = dy * ir; g.uz = dz * ir;
    g.d = masked ? -1.f : r;
    g.dd = (masked || !w) ? 0.f : g.ux * g.wx + g.uy * g.wy + g.uz * g.wz;
    return g;
}

template <bool MASK, bool DIAG>
__global__ void edge_geom_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                 const int64_t* __restrict__ nbr, const float* __restrict__ off, long long E,
                                 float* __restrict__ d, float* __restrict__ uhat, float* __restrict__ dd,
                                 float* __restrict__ ddel, MdgCell cell, float rc2, float* __restrict__ zero = nullptr,
                                 long long zero_n = 0) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    // (mdg_edge_geom_prepare: the per-edge accumulators of the reverse sweeps are cleared by the same launch)
    for (long long z = e; z < zero_n; z += (long long)gridDim.x * blockDim.x) zero[z] = 0.f;
    if (e >= E) return;
    const EdgeGeo g = edge_geometry<MASK, DIAG>(x, w, nbr, off, e, cell, rc2);
    d[e] = g.d;
    uhat[3 * e] = g.ux; uhat[3 * e + 1] = g.uy; uhat[3 * e + 2] = g.uz;
    if (w) {
        dd[e] = g.dd;
        ddel[3 * e] = g.wx; ddel[3 * e + 1] = g.wy; ddel[3 * e + 2] = g.wz;
    }
}

// F_n = -sum_slots sgn dU/dd uhat ;  (d(w.F)/dx)_n = -sum_slots sgn [ d_b uhat + dd_b / d (ddel - dd uhat) ]
// (sgn = +1 when n is the first atom of the edge).  16 lanes per atom, fixed combine order.
__global__ void edge_geom_bwd_kernel(const float* __restrict__ d_b, const float* __restrict__ dd_b,
                                     const float* __restrict__ d, const float* __restrict__ dd,
                                     const float* __restrict__ uhat, const float* __restrict__ ddel,
                                     const int32_t* __restrict__ col, const int32_t* __restrict__ eid,
                                     const int32_t* __restrict__ cnt, int N, int max_nbr, float* __restrict__ force,
                                     float* __restrict__ dwf) {
    const int n = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, sub = threadIdx.x & 15;
    float fx = 0.f, fy = 0.f, fz = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
    if (n < N) {
        const int m = cnt[n];
        const size_t row = (size_t)n * max_nbr;
        for (int k = sub; k < m; k += 16) {
            const int e = eid[row + k];
            const float sgn = col[row + k] > n ? 1.f : -1.f;
            const float ux = uhat[3 * e], uy = uhat[3 * e + 1], uz = uhat[3 * e + 2];
            const float a = sgn * dd_b[e];
            fx -= a * ux; fy -= a * uy; fz -= a * uz;
            if (d_b) {
                const float b = sgn * d_b[e], c = a / d[e], t = dd[e];
                gx -= b * ux + c * (ddel[3 * e] - t * ux);
                gy -= b * uy + c * (ddel[3 * e + 1] - t * uy);
                gz -= b * uz + c * (ddel[3 * e + 2] - t * uz);
            }
        }
    }
    fx = group_sum<16>(fx); fy = group_sum<16>(fy); fz = group_sum<16>(fz);
    if (d_b) { gx = group_sum<16>(gx); gy = group_sum<16>(gy); gz = group_sum<16>(gz); }
    if (n < N && sub == 0) {
        force[3 * n] = fx; force[3 * n + 1] = fy; force[3 * n + 2] = fz;
        if (d_b) { dwf[3 * n] = gx; dwf[3 * n + 1] = gy; dwf[3 * n + 2] = gz; }
    }
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Persistent grids are sized to ONE resident round: workgroups per CU from the occupancy API (registers + LDS of
// this instantiation) times the CU count.  (768 workgroups on 256 CUs at 2 resident per CU ran as a full round
// plus a half-empty one: -25 %.)  Cached per kernel instantiation and LDS size.
// the f32 filter sweeps as six bf16 piece products per operand pair (cfconv_fwd_x6_kernel): MDG_F32_X6=0 keeps the
// v_mfma_f32_16x16x4_f32 kernels
bool fwd_x6() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("MDG_F32_X6"); on = !(e && e[0] == '0'); }
    return on != 0;
}

// the SPLIT forward sweep (one atom per workgroup): systems whose four-atom workgroups would leave most CUs idle
// (MDG_FWD_SPLIT_ATOMS: the largest such system; 0: never)
bool fwd_split(int n_atoms) {
    static int lim = -1;
    if (lim < 0) { const char* e = getenv("MDG_FWD_SPLIT_ATOMS"); lim = e ? atoi(e) : 512; }
    return n_atoms <= lim;
}

template <typename K>
int resident_blocks(K kernel, size_t lds, int cap = 4) {
    struct Entry { const void* k; size_t l; int n; };
    static thread_local Entry cache[192];
    static thread_local int used = 0;
    for (int i = 0; i < used; ++i)
        if (cache[i].k == (const void*)kernel && cache[i].l == lds) return cache[i].n;
    int per_cu = 0, dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
        cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_cu > cap) per_cu = cap;
    const int n = per_cu * cus;
    if (used < 192) cache[used++] = Entry{(const void*)kernel, lds, n};
    return n;
}

int shape_ok(const MdgFilterNet* net, int& GP, int& FT) {
    MDG_CHECK_ARG(net && net->mu && net->coef && net->W1 && net->b1 && net->W2 && net->b2, "cfconv: null filter weights");
    const int G = net->n_gauss, F = net->n_filters;
    MDG_CHECK_ARG(G >= 1 && G <= 64, "cfconv: 1 <= n_gaussians <= 64 (got %d)", G);
    MDG_CHECK_ARG(mdg_cfconv_supported(G, F),
                  "cfconv: n_filters must be a multiple of 4 up to 64, of 8 up to 128, or of 128 up to 512 (got %d)", F);
    GP = G <= 32 ? 32 : 64;
    FT = F <= 64 ? 4 : 8;
    return MDG_OK;
}

// Layers wider than 128 filters run as chunks of 128: W(d) is a per-filter function, so chunk c of the output
// (columns 128 c ...) needs rows 128 c ... of W2 / b2 and the same columns of the node rows -- one launch per chunk with
// offset pointers and the full row stride (layer 1 of the filter network is recomputed per chunk); the adjoints are linear
// in the filter adjoint rows, so the reverse sweep adds the chunks up (d_b / dd_b in place, gW1 / gb1 in the reduce).
constexpr int F_CHUNK = 128;

FilterDev dev_of(const MdgFilterNet* net, int f0) {
    const int F = net->n_filters, fc = F - f0 < F_CHUNK ? F - f0 : F_CHUNK;
    return FilterDev{net->mu, net->coef, net->W1, net->b1, net->W2 + (size_t)f0 * net->n_gauss, net->b2 + f0, net->n_gauss, fc, F, F};
}

inline const float* at_col(const float* p, int f0) { return p ? p + f0 : nullptr; }
inline float* at_col(float* p, int f0) { return p ? p + f0 : nullptr; }
// column f0 of a bf16 mirror that travels through the kernels' `const float*` argument slots
inline const float* at_col16(const float* p, int f0) {
    return p ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(p) + f0) : nullptr;
}
inline bool rows16_ok(const MdgFilterNet* net) { return net->n_filters > 64 && net->n_filters % 8 == 0; }

int bwd_blocks(long long n_edges, bool theta) {
    const long long tiles = (n_edges + 63) / 64;
#ifndef MDG_BWD_THETA_BLOCKS
#define MDG_BWD_THETA_BLOCKS 512
#endif
    const long long want = theta ? MDG_BWD_THETA_BLOCKS : 768;
    return (int)(tiles < want ? (tiles > 0 ? tiles : 1) : want);
}

}  // namespace

extern "C" int mdg_cfconv_bias_column(int n_gauss) {
    return n_gauss >= 1 && n_gauss <= 64 && n_gauss != 32 && n_gauss != 64;       // a spare padded column (B2COL_BIAS)
}

extern "C" int mdg_cfconv_supported(int n_gauss, int n_filters) {
    if (n_gauss < 1 || n_gauss > 64 || n_filters < 4) return 0;
    if (n_filters <= 128) return n_filters % (n_filters <= 64 ? 4 : 8) == 0;
    return n_filters <= 512 && n_filters % 128 == 0;          // chunks of 128 filters per launch (csrc/cfconv_fused.hip)
}

extern "C" int mdg_edge_geom(const float* x, const float* w, const int64_t* nbr, const float* offsets, int64_t n_edges,
                             float* d, float* uhat, float* dd, float* ddel, void* stream) {
    MDG_CHECK_ARG(n_edges >= 0, "edge_geom: bad size");
    if (n_edges == 0) return MDG_OK;
    MDG_CHECK_ARG(x && nbr && offsets && d && uhat && (!w || (dd && ddel)), "edge_geom: null buffer");
    hipLaunchKernelGGL((edge_geom_kernel<false, true>), dim3((unsigned)((n_edges + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, w,
                       nbr, offsets, (long long)n_edges, d, uhat, dd, ddel, MdgCell{}, 0.f);
    MDG_CHECK_LAUNCH("edge_geom_kernel");
    return MDG_OK;
}

extern "C" int mdg_edge_geom_masked(const float* x, const float* w, const int64_t* nbr, const float* offsets, int64_t n_edges,
                                    const MdgCell* cell, float cutoff, float* d, float* uhat, float* dd, float* ddel,
                                    void* stream) {
    MDG_CHECK_ARG(n_edges >= 0 && cell && cutoff > 0.f, "edge_geom_masked: bad arguments");
    if (n_edges == 0) return MDG_OK;
    MDG_CHECK_ARG(x && nbr && offsets && d && uhat && (!w || (dd && ddel)), "edge_geom_masked: null buffer");
    const dim3 grid((unsigned)((n_edges + 255) / 256));
    if (cell->diag)
        hipLaunchKernelGGL((edge_geom_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, x, w, nbr, offsets,
                           (long long)n_edges, d, uhat, dd, ddel, *cell, cutoff * cutoff);
    else
        hipLaunchKernelGGL((edge_geom_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, x, w, nbr, offsets,
                           (long long)n_edges, d, uhat, dd, ddel, *cell, cutoff * cutoff);
    MDG_CHECK_LAUNCH("edge_geom_kernel");
    return MDG_OK;
}

// mdg_edge_geom (cell == NULL) / mdg_edge_geom_masked in one launch with the zero fill of `zero_n` floats at `zero` (the per-edge
// accumulators d_b / dd_b the reverse sweeps of the same evaluation add into): one launch less per evaluation.
extern "C" int mdg_edge_geom_prepare(const float* x, const float* w, const int64_t* nbr, const float* offsets, int64_t n_edges,
                                     const MdgCell* cell, float cutoff, float* d, float* uhat, float* dd, float* ddel,
                                     float* zero, int64_t zero_n, void* stream) {
    MDG_CHECK_ARG(n_edges >= 0 && zero_n >= 0 && (!cell || cutoff > 0.f), "edge_geom_prepare: bad arguments");
    if (n_edges == 0 && zero_n == 0) return MDG_OK;
    MDG_CHECK_ARG((n_edges == 0 || (x && nbr && offsets && d && uhat)) && (!w || (dd && ddel)) && (zero_n == 0 || zero),
                  "edge_geom_prepare: null buffer");
    const long long work = n_edges > 0 ? n_edges : 256;
    const dim3 grid((unsigned)((work + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    if (!cell)
        hipLaunchKernelGGL((edge_geom_kernel<false, true>), grid, dim3(256), 0, st, x, w, nbr, offsets, (long long)n_edges, d, uhat,
                           dd, ddel, MdgCell{}, 0.f, zero, (long long)zero_n);
    else if (cell->diag)
        hipLaunchKernelGGL((edge_geom_kernel<true, true>), grid, dim3(256), 0, st, x, w, nbr, offsets, (long long)n_edges, d, uhat,
                           dd, ddel, *cell, cutoff * cutoff, zero, (long long)zero_n);
    else
        hipLaunchKernelGGL((edge_geom_kernel<true, false>), grid, dim3(256), 0, st, x, w, nbr, offsets, (long long)n_edges, d, uhat,
                           dd, ddel, *cell, cutoff * cutoff, zero, (long long)zero_n);
    MDG_CHECK_LAUNCH("edge_geom_kernel");
    return MDG_OK;
}

extern "C" int mdg_edge_geom_bwd(const float* d_b, const float* dd_b, const float* d, const float* dd, const float* uhat,
                                 const float* ddel, const int32_t* col, const int32_t* eid, const int32_t* cnt,
                                 int n_atoms, int max_nbr, float* force, float* dwf, void* stream) {
    MDG_CHECK_ARG(dd_b && uhat && col && eid && cnt && force && n_atoms > 0, "edge_geom_bwd: bad arguments");
    MDG_CHECK_ARG(!d_b || (d && dd && ddel && dwf), "edge_geom_bwd: the second-order output needs d, dd, ddel");
    hipLaunchKernelGGL(edge_geom_bwd_kernel, dim3((n_atoms * 16 + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_b, dd_b,
                       d, dd, uhat, ddel, col, eid, cnt, n_atoms, max_nbr, force, dwf);
    MDG_CHECK_LAUNCH("edge_geom_bwd_kernel");
    return MDG_OK;
}

extern "C" int mdg_cfconv_fwd(const MdgFilterNet* net, const float* d, const float* dd, const float* h, const float* hd,
                              const int32_t* col, const int32_t* eid, const int32_t* cnt, int n_atoms, int max_nbr,
                              float* m, float* md, float* hsum, float* hdsum, void* stream) {
    int GP, FT;
    int rc = shape_ok(net, GP, FT);
    if (rc) return rc;
    MDG_CHECK_ARG(d && h && col && eid && cnt && m && n_atoms > 0 && max_nbr > 0, "cfconv_fwd: bad arguments");
    const bool tangent = dd != nullptr;
    MDG_CHECK_ARG(!tangent || md, "cfconv_fwd: the tangent sweep needs md");
    MDG_CHECK_ARG((long long)n_atoms * net->n_filters < (1LL << 30), "cfconv_fwd: n_atoms x n_filters must stay below 2^30");
    MDG_CHECK_ARG(tangent || (!hd && !md && !hdsum), "cfconv_fwd: tangent buffers without dd");
    MDG_CHECK_ARG(aligned16(h) && aligned16(hd) && aligned16(m) && aligned16(md) && aligned16(hsum) && aligned16(hdsum),
                  "cfconv_fwd: node feature matrices must be 16-byte aligned");
    const int most = (n_atoms + 3) / 4;
    const bool split = fwd_split(n_atoms);
    hipStream_t st = (hipStream_t)stream;
    for (int f0 = 0; f0 < net->n_filters; f0 += F_CHUNK) {
    FwdArgs a{dev_of(net, f0), d, dd, at_col(h, f0), at_col(hd, f0), col, eid, cnt, n_atoms, max_nbr, at_col(m, f0),
              at_col(md, f0), at_col(hsum, f0), at_col(hdsum, f0)};
    if (GP == 32 && fwd_x6()) {
        // n_gaussians <= 32: the f32-accurate sweep on the bf16 matrix pipe (cfconv_fwd_x6_kernel)
#define MDG_X6_3(FT_, T_, S_, H_, P_)                                                                              \
    do {                                                                                                           \
        const size_t lds = fwd_x6_lds_bytes<FT_>(T_);                                                              \
        if (P_) { hipLaunchKernelGGL((cfconv_fwd_x6_kernel<FT_, T_, S_, H_, P_>), dim3(n_atoms), dim3(256), lds, st, a); break; } \
        const int want = resident_blocks(cfconv_fwd_x6_kernel<FT_, T_, S_, H_, P_>, lds);                          \
        hipLaunchKernelGGL((cfconv_fwd_x6_kernel<FT_, T_, S_, H_, P_>), dim3(most < want ? most : want), dim3(256), lds, st, a); \
    } while (0)
#define MDG_X6_2(FT_, T_, S_, H_) do { if (split) MDG_X6_3(FT_, T_, S_, H_, true); else MDG_X6_3(FT_, T_, S_, H_, false); } while (0)
#define MDG_X6_1(FT_)                                                                                              \
    do {                                                                                                           \
        if (tangent && hd) { if (hsum) MDG_X6_2(FT_, true, true, true); else MDG_X6_2(FT_, true, false, true); }   \
        else if (tangent) { if (hsum) MDG_X6_2(FT_, true, true, false); else MDG_X6_2(FT_, true, false, false); }  \
        else { if (hsum) MDG_X6_2(FT_, false, true, false); else MDG_X6_2(FT_, false, false, false); }             \
    } while (0)
        if (FT == 4) MDG_X6_1(4); else MDG_X6_1(8);
#undef MDG_X6_1
#undef MDG_X6_2
#undef MDG_X6_3
        continue;
    }
#define MDG_FWD1(GP_, FT_, T_, S_)                                                                                 \
    do {                                                                                                           \
        const size_t lds = fwd_lds_bytes<GP_, FT_>(T_);                                                            \
        if (split) {            /* few atoms: one atom per workgroup, its tiles dealt to the four waves */          \
            hipLaunchKernelGGL((cfconv_fwd_kernel<GP_, FT_, T_, S_, true>), dim3(n_atoms), dim3(256), lds, st, a); \
            break;                                                                                                 \
        }                                                                                                          \
        const int want = resident_blocks(cfconv_fwd_kernel<GP_, FT_, T_, S_>, lds);                                \
        hipLaunchKernelGGL((cfconv_fwd_kernel<GP_, FT_, T_, S_>), dim3(most < want ? most : want), dim3(256), lds, st, a); \
    } while (0)
#define MDG_FWD(GP_, FT_)                                                                                          \
    do {                                                                                                           \
        if (tangent) { if (hsum) MDG_FWD1(GP_, FT_, true, true); else MDG_FWD1(GP_, FT_, true, false); }           \
        else { if (hsum) MDG_FWD1(GP_, FT_, false, true); else MDG_FWD1(GP_, FT_, false, false); }                 \
    } while (0)
    if (GP == 32 && FT == 4) MDG_FWD(32, 4);
    else if (GP == 32) MDG_FWD(32, 8);
    else if (FT == 4) MDG_FWD(64, 4);
    else MDG_FWD(64, 8);
#undef MDG_FWD
#undef MDG_FWD1
    }
    MDG_CHECK_LAUNCH("cfconv_fwd_kernel");
    return MDG_OK;
}

namespace {
int cfconv_fwd_bf16_impl(const MdgFilterNet* net, const float* d, const float* dd, const float* h,
                         const float* hd, const int32_t* col, const int32_t* eid, const int32_t* cnt,
                         int n_atoms, int max_nbr, float* m, float* md, float* hsum, float* hdsum,
                         void* stream, bool rows16, const uint16_t* st_s = nullptr, const uint16_t* st_sd = nullptr) {
    int GP, FT;
    int rc = shape_ok(net, GP, FT);
    if (rc) return rc;
    const bool stash = st_s != nullptr;
    MDG_CHECK_ARG((d || stash) && h && col && eid && cnt && m && n_atoms > 0 && max_nbr > 0, "cfconv_fwd_bf16: bad arguments");
    const bool tangent = stash ? st_sd != nullptr : dd != nullptr;
    MDG_CHECK_ARG(!stash || (!hsum && !hdsum && (d || net->n_gauss + 2 <= GP)),
                  "cfconv_fwd_stashed: no neighbour sums; the distances are needed when the bias does not ride in the k columns");
    MDG_CHECK_ARG(!tangent || md, "cfconv_fwd_bf16: the tangent sweep needs md");
    MDG_CHECK_ARG((long long)n_atoms * net->n_filters < (1LL << 30), "cfconv_fwd_bf16: n_atoms x n_filters must stay below 2^30");
    MDG_CHECK_ARG(tangent || (!hd && !md && !hdsum), "cfconv_fwd_bf16: tangent buffers without dd");
    MDG_CHECK_ARG(aligned16(h) && aligned16(hd) && aligned16(m) && aligned16(md) && aligned16(hsum) && aligned16(hdsum),
                  "cfconv_fwd_bf16: node feature matrices must be 16-byte aligned");
    MDG_CHECK_ARG(!rows16 || rows16_ok(net), "cfconv_fwd_rows16: bf16 node rows need n_filters > 64, a multiple of 8 (got %d)",
                  net->n_filters);
    const int most = (n_atoms + 3) / 4;
    hipStream_t st = (hipStream_t)stream;
    for (int f0 = 0; f0 < net->n_filters; f0 += F_CHUNK) {
    FwdArgs a{dev_of(net, f0), d, dd, rows16 ? at_col16(h, f0) : at_col(h, f0), rows16 ? at_col16(hd, f0) : at_col(hd, f0), col,
              eid, cnt, n_atoms, max_nbr, at_col(m, f0), at_col(md, f0), at_col(hsum, f0), at_col(hdsum, f0), st_s, st_sd};
#define MDG_FWDB4(GP_, FT_, T_, S_, H_, B_, R_, X_)                                                                \
    do {                                                                                                           \
        const size_t lds = fwd_bf16_lds_bytes<GP_, FT_>(T_);                                                       \
        const int want = resident_blocks(cfconv_fwd_bf16_kernel<GP_, FT_, T_, S_, H_, B_, R_, X_>, lds, X_ ? 6 : 4); \
        hipLaunchKernelGGL((cfconv_fwd_bf16_kernel<GP_, FT_, T_, S_, H_, B_, R_, X_>), dim3(most < want ? most : want), dim3(256), lds, st, a); \
    } while (0)
#define MDG_FWDB3(GP_, FT_, T_, S_, H_, B_, R_)                                                                    \
    do {                                                                                                           \
        if (stash) MDG_FWDB4(GP_, FT_, T_, false, H_, B_, R_, (!(S_)));   /* (S_ = true never comes with a stash) */ \
        else MDG_FWDB4(GP_, FT_, T_, S_, H_, B_, R_, false);                                                       \
    } while (0)
#define MDG_FWDB2(GP_, FT_, T_, S_, H_, B_)                                                                        \
    do {                                                                                                           \
        if (rows16) MDG_FWDB3(GP_, FT_, T_, S_, H_, B_, (FT_ == 8));   /* (FT = 4 never gets here: rows16_ok) */   \
        else MDG_FWDB3(GP_, FT_, T_, S_, H_, B_, false);                                                           \
    } while (0)
#define MDG_FWDB1(GP_, FT_, T_, S_, H_)                                                                            \
    do { if (net->n_gauss + 2 <= GP_) MDG_FWDB2(GP_, FT_, T_, S_, H_, true); else MDG_FWDB2(GP_, FT_, T_, S_, H_, false); } while (0)
#define MDG_FWDB(GP_, FT_)                                                                                         \
    do {                                                                                                           \
        if (tangent && hd) { if (hsum) MDG_FWDB1(GP_, FT_, true, true, true); else MDG_FWDB1(GP_, FT_, true, false, true); }   \
        else if (tangent) { if (hsum) MDG_FWDB1(GP_, FT_, true, true, false); else MDG_FWDB1(GP_, FT_, true, false, false); } \
        else { if (hsum) MDG_FWDB1(GP_, FT_, false, true, false); else MDG_FWDB1(GP_, FT_, false, false, false); }             \
    } while (0)
    if (GP == 32 && FT == 4) MDG_FWDB(32, 4);
    else if (GP == 32) MDG_FWDB(32, 8);
    else if (FT == 4) MDG_FWDB(64, 4);
    else MDG_FWDB(64, 8);
#undef MDG_FWDB4
#undef MDG_FWDB3
#undef MDG_FWDB2
#undef MDG_FWDB
#undef MDG_FWDB1
    }
    MDG_CHECK_LAUNCH("cfconv_fwd_bf16_kernel");
    return MDG_OK;
}
}  // namespace

extern "C" int mdg_cfconv_fwd_bf16(const MdgFilterNet* net, const float* d, const float* dd, const float* h,
                                   const float* hd, const int32_t* col, const int32_t* eid, const int32_t* cnt,
                                   int n_atoms, int max_nbr, float* m, float* md, float* hsum, float* hdsum,
                                   void* stream) {
    return cfconv_fwd_bf16_impl(net, d, dd, h, hd, col, eid, cnt, n_atoms, max_nbr, m, md, hsum, hdsum, stream, false);
}

// mdg_cfconv_fwd_bf16 over bf16 MIRRORS of the gathered node matrices: h16 / hd16 are [n_atoms, n_filters] bf16 (dense rows);
// the outputs stay f32.  A precision option of its own (the node rows lose 16 mantissa bits before they are multiplied), see
// include/mdgrad_hip.h.
extern "C" int mdg_cfconv_fwd_rows16(const MdgFilterNet* net, const float* d, const float* dd, const uint16_t* h16,
                                     const uint16_t* hd16, const int32_t* col, const int32_t* eid, const int32_t* cnt,
                                     int n_atoms, int max_nbr, float* m, float* md, float* hsum, float* hdsum,
                                     void* stream) {
    return cfconv_fwd_bf16_impl(net, d, dd, reinterpret_cast<const float*>(h16), reinterpret_cast<const float*>(hd16), col, eid,
                                cnt, n_atoms, max_nbr, m, md, hsum, hdsum, stream, true);
}

// The first Dense layer of the filter network once per undirected edge: st_s (and, with dd, st_sd) [n_edges][GP] bf16,
// GP = 32 for n_gaussians <= 32, else 64 (mdg_cfconv_stash_width).  See filter_stash_kernel.
extern "C" int mdg_cfconv_stash_width(int n_gauss) { return n_gauss <= 32 ? 32 : 64; }

extern "C" int mdg_cfconv_filter_stash(const MdgFilterNet* net, const float* d, const float* dd, int64_t n_edges,
                                       const int32_t* n_valid, uint16_t* st_s, uint16_t* st_sd, void* stream) {
    int GP, FT;
    int rc = shape_ok(net, GP, FT);
    if (rc) return rc;
    MDG_CHECK_ARG(d && st_s && n_edges >= 0 && (dd == nullptr) == (st_sd == nullptr), "cfconv_filter_stash: bad arguments");
    MDG_CHECK_ARG(aligned16(st_s) && aligned16(st_sd), "cfconv_filter_stash: the stash must be 16-byte aligned");
    MDG_CHECK_ARG(n_edges * (long long)GP < (1LL << 32), "cfconv_filter_stash: n_edges x %d must stay below 2^32", GP);
    if (n_edges == 0) return MDG_OK;
    const bool tangent = dd != nullptr, biask = net->n_gauss + 2 <= GP;
    StashArgs a{dev_of(net, 0), d, dd, (long long)n_edges, n_valid, st_s, st_sd};
    const long long tiles = (n_edges + 63) / 64;
    const int nb = (int)(tiles < 2048 ? tiles : 2048);
    hipStream_t st = (hipStream_t)stream;
#define MDG_STASH(GP_)                                                                                             \
    do {                                                                                                           \
        const size_t lds = stash_lds_bytes<GP_>(tangent);                                                          \
        if (tangent) { if (biask) hipLaunchKernelGGL((filter_stash_kernel<GP_, true, true>), dim3(nb), dim3(256), lds, st, a);   \
                       else hipLaunchKernelGGL((filter_stash_kernel<GP_, true, false>), dim3(nb), dim3(256), lds, st, a); }      \
        else { if (biask) hipLaunchKernelGGL((filter_stash_kernel<GP_, false, true>), dim3(nb), dim3(256), lds, st, a);          \
               else hipLaunchKernelGGL((filter_stash_kernel<GP_, false, false>), dim3(nb), dim3(256), lds, st, a); }             \
    } while (0)
    if (GP == 32) MDG_STASH(32); else MDG_STASH(64);
#undef MDG_STASH
    MDG_CHECK_LAUNCH("filter_stash_kernel");
    return MDG_OK;
}

// mdg_cfconv_fwd_bf16 / mdg_cfconv_fwd_rows16 (rows16 != 0: h, hd are bf16 mirrors) with the second layer's operands read
// from the stash of mdg_cfconv_filter_stash: tangent sweep iff st_sd (and md) are given.  d: the edge distances, needed only
// when the bias does not ride in the k columns (n_gaussians + 2 > stash width); NULL otherwise.  Bitwise the outputs of the
// recomputing entry points.
extern "C" int mdg_cfconv_fwd_stashed(const MdgFilterNet* net, const uint16_t* st_s, const uint16_t* st_sd, const float* d,
                                      const void* h, const void* hd, const int32_t* col, const int32_t* eid, const int32_t* cnt,
                                      int n_atoms, int max_nbr, float* m, float* md, int rows16, void* stream) {
    MDG_CHECK_ARG(st_s && (st_sd != nullptr) == (md != nullptr) && (st_sd || !hd), "cfconv_fwd_stashed: the tangent operands come together");
    return cfconv_fwd_bf16_impl(net, d, nullptr, static_cast<const float*>(h), static_cast<const float*>(hd), col, eid, cnt, n_atoms,
                                max_nbr, m, md, nullptr, nullptr, stream, rows16 != 0, st_s, st_sd);
}

extern "C" int mdg_cfconv_rows16_supported(int n_gauss, int n_filters) {
    return mdg_cfconv_supported(n_gauss, n_filters) && n_filters > 64 && n_filters % 8 == 0;
}

extern "C" int64_t mdg_cfconv_bwd_workspace(int n_gauss, int n_filters, int64_t n_edges) {
    if (!mdg_cfconv_supported(n_gauss, n_filters) || n_edges <= 0) return 0;
    const int GP = n_gauss <= 32 ? 32 : 64, FP = n_filters <= 64 ? 64 : 128;
    return (int64_t)bwd_blocks(n_edges, true) * (GP * GP + 3 * GP + FP * GP);
}

namespace {

// the rows16 instantiation behind the launch macro's four-parameter kernel name (FT = 4 never gets here: rows16_ok)
template <int GP, int FT, bool DUAL, bool THETA>
constexpr auto bwd_r16_kernel = cfconv_bwd_bf16_kernel<GP, FT, DUAL, THETA, (FT == 8)>;

int cfconv_bwd_impl(const MdgFilterNet* net, const float* d, const float* dd, const int64_t* nbr,
                    int64_t n_edges, const float* h, const float* hd, const float* mb, const float* mdb,
                    float* d_b, float* dd_b, float* gW1, float* gb1, float* gW2, float* workspace,
                    const int32_t* n_valid, void* stream, bool bf16, float* gmu = nullptr, float* gcoef = nullptr,
                    bool rows16 = false, float* gb2 = nullptr) {
    int GP, FT;
    int rc = shape_ok(net, GP, FT);
    if (rc) return rc;
    MDG_CHECK_ARG(n_edges >= 0, "cfconv_bwd: bad size");
    const bool dual = mb != nullptr, theta = gW1 != nullptr;
    MDG_CHECK_ARG(!theta || (dual && gb1 && gW2), "cfconv_bwd: parameter gradients come from the dual sweep");
    hipStream_t st = (hipStream_t)stream;
    if (n_edges == 0) {
        if (theta) {
            const int G = net->n_gauss, F = net->n_filters;
            MDG_HIP(hipMemsetAsync(gW1, 0, sizeof(float) * G * G, st));
            MDG_HIP(hipMemsetAsync(gb1, 0, sizeof(float) * G, st));
            MDG_HIP(hipMemsetAsync(gW2, 0, sizeof(float) * F * G, st));
            if (gmu) MDG_HIP(hipMemsetAsync(gmu, 0, sizeof(float) * G, st));
            if (gcoef) MDG_HIP(hipMemsetAsync(gcoef, 0, sizeof(float) * G, st));
            if (gb2) MDG_HIP(hipMemsetAsync(gb2, 0, sizeof(float) * F, st));
        }
        return MDG_OK;
    }
    MDG_CHECK_ARG(!gb2 || (theta && mdg_cfconv_bias_column(net->n_gauss)),
                  "cfconv_bwd: the bias-column gradient needs the parameter gradients and n_gaussians below the padded width (got %d)",
                  net->n_gauss);
    MDG_CHECK_ARG(d && nbr && h && mdb && dd_b, "cfconv_bwd: null buffer");
    MDG_CHECK_ARG(!dual || (dd && d_b), "cfconv_bwd: the dual sweep needs dd and d_b");
    MDG_CHECK_ARG(dual || !hd, "cfconv_bwd: hd without the dual sweep");
    MDG_CHECK_ARG(!theta || workspace, "cfconv_bwd: workspace missing");
    MDG_CHECK_ARG(aligned16(h) && aligned16(hd) && aligned16(mb) && aligned16(mdb) && aligned16(nbr),
                  "cfconv_bwd: node feature matrices and the pair list must be 16-byte aligned");
    MDG_CHECK_ARG(!rows16 || (bf16 && rows16_ok(net)),
                  "cfconv_bwd_rows16: bf16 node rows need n_filters > 64, a multiple of 8 (got %d)", net->n_filters);
    const long long tiles64 = (n_edges + 63) / 64;
    for (int f0 = 0; f0 < net->n_filters; f0 += F_CHUNK) {
    BwdArgs a{dev_of(net, f0), d, dd, nbr, (long long)n_edges, rows16 ? at_col16(h, f0) : at_col(h, f0),
              rows16 ? at_col16(hd, f0) : at_col(hd, f0), rows16 ? at_col16(mb, f0) : at_col(mb, f0),
              rows16 ? at_col16(mdb, f0) : at_col(mdb, f0), d_b, dd_b, workspace, n_valid};
    int nb = bwd_blocks(n_edges, theta);             // (theta: the workspace holds one record per workgroup, <= 512)
#define MDG_BWD_K(K_, L_, GP_, FT_)                                                                                \
    do {                                                                                                           \
        if (theta) {                                                                                               \
            const size_t lds = L_<GP_, FT_>(true);                                                                 \
            const int want = resident_blocks(K_<GP_, FT_, true, true>, lds);                                       \
            if (want < nb) nb = want;                                                                              \
            hipLaunchKernelGGL((K_<GP_, FT_, true, true>), dim3(nb), dim3(256), lds, st, a);                       \
        } else if (dual) {                                                                                         \
            const size_t lds = L_<GP_, FT_>(false);                                                                \
            const int want = resident_blocks(K_<GP_, FT_, true, false>, lds);                                      \
            nb = (int)(tiles64 < want ? tiles64 : want);                                                           \
            hipLaunchKernelGGL((K_<GP_, FT_, true, false>), dim3(nb), dim3(256), lds, st, a);                      \
        } else {                                                                                                   \
            const size_t lds = L_<GP_, FT_>(false);                                                                \
            const int want = resident_blocks(K_<GP_, FT_, false, false>, lds);                                     \
            nb = (int)(tiles64 < want ? tiles64 : want);                                                           \
            hipLaunchKernelGGL((K_<GP_, FT_, false, false>), dim3(nb), dim3(256), lds, st, a);                     \
        }                                                                                                          \
    } while (0)
#define MDG_BWD(GP_, FT_)                                                                                          \
    do {                                                                                                           \
        if (rows16) MDG_BWD_K(bwd_r16_kernel, bwd_bf16_lds_bytes, GP_, FT_);                                       \
        else if (bf16) MDG_BWD_K(cfconv_bwd_bf16_kernel, bwd_bf16_lds_bytes, GP_, FT_);                            \
        else MDG_BWD_K(cfconv_bwd_kernel, bwd_lds_bytes, GP_, FT_);                                                \
    } while (0)
    if (GP == 32 && FT == 4) MDG_BWD(32, 4);
    else if (GP == 32) MDG_BWD(32, 8);
    else if (FT == 4) MDG_BWD(64, 4);
    else MDG_BWD(64, 8);
#undef MDG_BWD
#undef MDG_BWD_K
    MDG_CHECK_LAUNCH("cfconv_bwd_kernel");
    if (theta) {
        const int FP = 16 * FT, REC = GP * GP + 3 * GP + FP * GP;
        hipLaunchKernelGGL(cfconv_bwd_reduce_kernel, dim3((REC + 63) / 64), dim3(64 * RED_WAVES), 0, st, workspace, nb, GP, FP,
                           net->n_gauss, a.net.F, gW1, gb1, gW2 + (size_t)f0 * net->n_gauss, gmu, gcoef, f0 > 0 ? 1 : 0,
                           gb2 ? gb2 + f0 : nullptr);
        MDG_CHECK_LAUNCH("cfconv_bwd_reduce_kernel");
    }
    }
    return MDG_OK;
}

}  // namespace

extern "C" int mdg_cfconv_bwd(const MdgFilterNet* net, const float* d, const float* dd, const int64_t* nbr,
                              int64_t n_edges, const float* h, const float* hd, const float* mb, const float* mdb,
                              float* d_b, float* dd_b, float* gW1, float* gb1, float* gW2, float* workspace,
                              const int32_t* n_valid, void* stream) {
    return cfconv_bwd_impl(net, d, dd, nbr, n_edges, h, hd, mb, mdb, d_b, dd_b, gW1, gb1, gW2, workspace, n_valid, stream, false);
}

extern "C" int mdg_cfconv_bwd_bf16(const MdgFilterNet* net, const float* d, const float* dd, const int64_t* nbr,
                                   int64_t n_edges, const float* h, const float* hd, const float* mb, const float* mdb,
                                   float* d_b, float* dd_b, float* gW1, float* gb1, float* gW2, float* workspace,
                                   const int32_t* n_valid, void* stream) {
    return cfconv_bwd_impl(net, d, dd, nbr, n_edges, h, hd, mb, mdb, d_b, dd_b, gW1, gb1, gW2, workspace, n_valid, stream, true);
}


// mdg_cfconv_bwd / mdg_cfconv_bwd_bf16 (bf16 != 0) with the gradients of the Gaussian basis as well: gmu[G] = d/d(mu_k),
// gcoef[G] = d/d(coef_k) of the same scalar (nff/nn/layers.py:34-83 with trainable = True: `offsets` and `width` are
// parameters; coef = -0.5 / width^2 is chained by the caller).
extern "C" int mdg_cfconv_bwd_smear(const MdgFilterNet* net, const float* d, const float* dd, const int64_t* nbr,
                                    int64_t n_edges, const float* h, const float* hd, const float* mb, const float* mdb,
                                    float* d_b, float* dd_b, float* gW1, float* gb1, float* gW2, float* gmu, float* gcoef,
                                    float* workspace, const int32_t* n_valid, int bf16, void* stream) {
    MDG_CHECK_ARG(gW1 && gmu && gcoef, "cfconv_bwd_smear: the basis gradients come with the parameter gradients");
    return cfconv_bwd_impl(net, d, dd, nbr, n_edges, h, hd, mb, mdb, d_b, dd_b, gW1, gb1, gW2, workspace, n_valid, stream,
                           bf16 != 0, gmu, gcoef);
}

// mdg_cfconv_bwd_bf16 / mdg_cfconv_bwd_smear over bf16 MIRRORS of the four gathered node matrices ([n_atoms, n_filters] bf16,
// dense rows; hd16 may be null as hd may).  gmu / gcoef null: no basis gradients.  See mdg_cfconv_fwd_rows16.
extern "C" int mdg_cfconv_bwd_rows16(const MdgFilterNet* net, const float* d, const float* dd, const int64_t* nbr,
                                     int64_t n_edges, int n_atoms, const uint16_t* h16, const uint16_t* hd16,
                                     const uint16_t* mb16, const uint16_t* mdb16, float* d_b, float* dd_b, float* gW1,
                                     float* gb1, float* gW2, float* gmu, float* gcoef, float* workspace,
                                     const int32_t* n_valid, void* stream) {
    MDG_CHECK_ARG(net && n_atoms > 0 && (long long)n_atoms * net->n_filters < (1LL << 30),
                  "cfconv_bwd_rows16: n_atoms x n_filters must stay below 2^30");
    MDG_CHECK_ARG((gmu == nullptr) == (gcoef == nullptr) && (!gmu || gW1), "cfconv_bwd_rows16: the basis gradients come together");
    return cfconv_bwd_impl(net, d, dd, nbr, n_edges, reinterpret_cast<const float*>(h16), reinterpret_cast<const float*>(hd16),
                           reinterpret_cast<const float*>(mb16), reinterpret_cast<const float*>(mdb16), d_b, dd_b, gW1, gb1,
                           gW2, workspace, n_valid, stream, true, gmu, gcoef, true);
}

// [n_rows, n_cols] f32 rows (row stride src_stride floats) -> dense bf16 rows, round to nearest even: the mirror of a node
// matrix that was not produced by mdg_row_chain (which writes its mirrors itself).
namespace {
__global__ __launch_bounds__(256) void rows_to_bf16_kernel(const float* __restrict__ src, long long n_rows, int n_cols,
                                                           int src_stride, unsigned short* __restrict__ dst) {
    const int per = n_cols / 4;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_rows * per) return;
    const long long r = t / per;
    const int c = (int)(t % per) * 4;
    const float4 v = *reinterpret_cast<const float4*>(src + r * src_stride + c);
    const u32x2v o = {cvt_pk_bf16(v.x, v.y), cvt_pk_bf16(v.z, v.w)};
    *reinterpret_cast<u32x2v*>(dst + r * n_cols + c) = o;
}
}  // namespace

extern "C" int mdg_rows_to_bf16(const float* src, int64_t n_rows, int n_cols, int src_stride, uint16_t* dst, void* stream) {
    MDG_CHECK_ARG(src && dst && n_rows >= 0 && n_cols > 0 && n_cols % 4 == 0 && src_stride >= n_cols && src_stride % 4 == 0,
                  "rows_to_bf16: bad arguments (columns and stride in multiples of 4)");
    MDG_CHECK_ARG(aligned16(src) && (((uintptr_t)dst) & 7) == 0, "rows_to_bf16: unaligned buffers");
    if (n_rows == 0) return MDG_OK;
    const long long work = n_rows * (n_cols / 4);
    hipLaunchKernelGGL(rows_to_bf16_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                       (long long)n_rows, n_cols, src_stride, dst);
    MDG_CHECK_LAUNCH("rows_to_bf16_kernel");
    return MDG_OK;
}

// The reverse sweep with parameter gradients, every option in one entry: flags = MDG_CFCONV_BF16 (bf16 MFMA operands) |
// MDG_CFCONV_ROWS16 (with BF16: h, hd, mb, mdb point at bf16 mirrors, see mdg_cfconv_fwd_rows16).  gb2[n_filters] (nullable,
// mdg_cfconv_bias_column(n_gauss) != 0): the gradient of the second filter layer's bias, sum_e Wb[e][f].  gmu / gcoef
// (nullable, together): the gradients of the Gaussian basis as in mdg_cfconv_bwd_smear.
extern "C" int mdg_cfconv_bwd_theta(const MdgFilterNet* net, const float* d, const float* dd, const int64_t* nbr,
                                    int64_t n_edges, int n_atoms, const void* h, const void* hd, const void* mb,
                                    const void* mdb, float* d_b, float* dd_b, float* gW1, float* gb1, float* gW2, float* gb2,
                                    float* gmu, float* gcoef, float* workspace, const int32_t* n_valid, int flags,
                                    void* stream) {
    const bool bf16 = (flags & MDG_CFCONV_BF16) != 0, rows16 = (flags & MDG_CFCONV_ROWS16) != 0;
    MDG_CHECK_ARG(net && gW1 && gb1 && gW2, "cfconv_bwd_theta: the parameter-gradient outputs are required");
    MDG_CHECK_ARG(!rows16 || (bf16 && n_atoms > 0 && (long long)n_atoms * net->n_filters < (1LL << 30)),
                  "cfconv_bwd_theta: bf16 node rows go with the bf16 kernels, n_atoms x n_filters below 2^30");
    MDG_CHECK_ARG((gmu == nullptr) == (gcoef == nullptr), "cfconv_bwd_theta: the basis gradients come together");
    return cfconv_bwd_impl(net, d, dd, nbr, n_edges, static_cast<const float*>(h), static_cast<const float*>(hd),
                           static_cast<const float*>(mb), static_cast<const float*>(mdb), d_b, dd_b, gW1, gb1, gW2, workspace,
                           n_valid, stream, bf16, gmu, gcoef, rows16, gb2);
}
