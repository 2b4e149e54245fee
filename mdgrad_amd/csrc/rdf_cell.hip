// K8 for large systems (N >= 2 048: BASELINE config #4's 4 096-atom liquid), straight from the cell bins:
//   raw[k] = sum over frames and pairs i<j of exp(coeff (d_ij - mu_k)^2)          (torchmd/observable.py:62-76)
// The reference builds a neighbour list per frame (generate_nbr_list, observable.py:64-66) and smears every listed
// distance (nff/nn/layers.py:19-29).  Materialising that list costs more than everything done with it (a rank-sorted
// row per atom: ~1 300 wave instructions per atom), so these kernels search and consume in one sweep:
//   * rdf_cell_bin_kernel   one workgroup per frame: LDS counts -> LDS scan -> positions sorted by (bin, atom index)
//                           as (x, y, z, index); the slot inside a bin is the atom's RANK by index among its bin mates,
//                           so the order -- and every floating-point sum taken over it -- is reproducible
//   * rdf_cell_fwd_kernel   a 16-lane row per atom (four atoms per wave) walks the 3 x 3 stencil columns (each column's
//                           three z-bins are ONE contiguous range of the sorted array), counts every pair once (neighbour index above the
//                           atom's) on the fine integer grid of the many-frame kernels (csrc/rdf.hip): LDS integer
//                           atomics, one flush per workgroup; mdg_rdf_fine_finish smears the counts onto the centres
//   * rdf_cell_bwd_kernel   the same sweep with the pair force of phi(d) = sum_k g_k exp(coeff (d - mu_k)^2) from its
//                           table (MDG_PAIR_TABLE, built by the caller from dL/d raw): dL/dx_i = sum_j phi'(d) D/d,
//                           every pair from both ends, no atomics
// Pair geometry as everywhere: reference minimum image (topology.py:59-64) and un-contracted d^2.
// Round 6: FINE z-bins (make_grid): up to 32 bins along z, a window of +- zw of them per stencil column -- still one contiguous
// range per column, 31 % fewer candidates at 4 096 atoms: forward 1.40 -> 1.25 ms, backward 2.46 -> 2.13 ms for 704 frames
// (tools/diag_rdf_zfine.py; counts identical, gradient to 2e-7).
//
// Round 5: COLUMN TILES in LDS (rdf_cell_fwd_tile_kernel / rdf_cell_bwd_tile_kernel).  The row sweeps above read every
// candidate of every atom from L2 (27 bins x ~19 atoms x 16 B per atom: 24 GB per backward call at 704 frames x 4 096 atoms --
// the kernels ran at the L2 -> L1 rate, 4x above their arithmetic).  A workgroup now owns one (bx, by) column of bins of one
// frame -- a contiguous range of the sorted array -- and stages the 3 x 3 columns around it (nine contiguous ranges, ~1 000
// atoms = 16 KB) in LDS once; every row then reads its candidates with conflict-free ds_read_b128 (16 consecutive float4 per
// row and pass).  Same candidates in the same per-lane order as the row sweeps: the backward sums are bit for bit the same,
// the forward counts are integers.  A tile whose nine columns exceed the staged capacity takes the row sweep from L2.
#include "common.hpp"

namespace {

constexpr int RC_MAX_CELLS = 4096;           // bins per frame (LDS scan)
constexpr int RC_THREADS = 1024;
constexpr int RC_MAX_ATOMS = RC_THREADS * 32;

constexpr int RC_MAX_NBZ = 32;               // bins along z at most (FINE z-bins, round 6)
struct CellGrid { int nb[3]; int ncell; int zw; };      // zw: half-width of the stencil along z, in bins

// MDG_RDF_CELL_ZFINE=0: z-bins of at least the cutoff like x and y (A/B measurements; round 5's grid)
bool zfine_enabled() {
    const char* e = getenv("MDG_RDF_CELL_ZFINE");
    return !(e && e[0] == '0');
}

// x, y: bins of at least the cutoff, a 3 x 3 stencil of bin COLUMNS.  z (round 6): as many bins as fit (<= RC_MAX_NBZ, <=
// RC_MAX_CELLS in all) and a stencil of +- zw of them, zw bins >= the cutoff: a column's window bz - zw .. bz + zw is still ONE
// contiguous range of the sorted array (plus one through the face), so the bookkeeping per row is unchanged, but the window is
// (2 zw + 1) / nbz of the column instead of three bins of >= the cutoff -- 5.8 instead of 8.5 length units at 4 096 atoms,
// cutoff 2.6: 31 % fewer candidates for the same pairs.
CellGrid make_grid(const MdgCell& c, float cutoff) {
    CellGrid g;
    for (int d = 0; d < 3; ++d) {
        int n = (int)floorf(c.h[4 * d] / cutoff);
        g.nb[d] = n > 16 ? 16 : n;                       // (bins wider than the cutoff are fine; 16^3 = RC_MAX_CELLS)
    }
    g.zw = 1;
    if (zfine_enabled() && g.nb[0] >= 3 && g.nb[1] >= 3 && g.nb[2] >= 3) {
        int nz = RC_MAX_CELLS / (g.nb[0] * g.nb[1]);
        if (nz > RC_MAX_NBZ) nz = RC_MAX_NBZ;
        if (nz > g.nb[2]) {
            const double bin = (double)c.h[8] / nz;
            int zw = (int)ceil((double)cutoff / bin);
            if (zw * bin < (double)cutoff * (1.0 + 1e-6)) ++zw;      // (a window that is the cutoff to the last bit: one more)
            // (the window and its part through the face must not overlap, and it has to pay: narrower than three coarse bins)
            if (2 * zw + 1 <= nz && (2 * zw + 1) * bin < 3.0 * (double)c.h[8] / g.nb[2]) { g.nb[2] = nz; g.zw = zw; }
        }
    }
    g.ncell = g.nb[0] * g.nb[1] * g.nb[2];
    return g;
}

__device__ __forceinline__ int bin_coord_c(float x, float inv, int nb) {
    float fr = x * inv;
    fr -= floorf(fr);
    const int b = (int)(fr * (float)nb);
    return b >= nb ? nb - 1 : (b < 0 ? 0 : b);
}

// ---------------------------------------------------------------------------------------------
template <int NA>
__global__ __launch_bounds__(RC_THREADS) void rdf_cell_bin_kernel(const float* __restrict__ xyz, int N, MdgCell cell,
                                                                  CellGrid g, int32_t* __restrict__ bstart,
                                                                  int32_t* __restrict__ tmp, float4* __restrict__ spos) {
    __shared__ int32_t start[RC_MAX_CELLS + 1];
    __shared__ int32_t tsum[RC_THREADS];
    const int f = blockIdx.x;
    const float* x = xyz + (size_t)f * N * 3;
    int32_t* tm = tmp + (size_t)f * N;
    float4* sp = spos + (size_t)f * N;
    for (int c = threadIdx.x; c <= RC_MAX_CELLS; c += RC_THREADS) start[c] = 0;
    __syncthreads();
    float px[NA], py[NA], pz[NA];
    int bsl[NA];
#pragma unroll
    for (int u = 0; u < NA; ++u) {
        const int a = threadIdx.x + u * RC_THREADS;
        if (a < N) {
            px[u] = x[3 * a]; py[u] = x[3 * a + 1]; pz[u] = x[3 * a + 2];
            const int bx = bin_coord_c(px[u], cell.inv[0], g.nb[0]);
            const int by = bin_coord_c(py[u], cell.inv[4], g.nb[1]);
            const int bz = bin_coord_c(pz[u], cell.inv[8], g.nb[2]);
            const int bin = (bx * g.nb[1] + by) * g.nb[2] + bz;
            bsl[u] = (atomicAdd(&start[bin], 1) << 12) | bin;          // (provisional slot: the atomic's order)
        }
    }
    __syncthreads();
    // exclusive scan of the bin counts: 4 consecutive bins per thread + scan of the thread totals (16 waves)
    int loc[4], tot = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) { loc[u] = start[threadIdx.x * 4 + u]; tot += loc[u]; }
    {
        int v = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(v, o, 64); if ((int)(threadIdx.x & 63) >= o) v += y; }
        tsum[threadIdx.x] = v;
    }
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += tsum[w * 64 + 63];
    int run = base + tsum[threadIdx.x] - tot;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) { start[threadIdx.x * 4 + u] = run; run += loc[u]; }
    if (threadIdx.x == RC_THREADS - 1) start[RC_MAX_CELLS] = run;
    __syncthreads();
    int32_t* bs = bstart + (size_t)f * (g.ncell + 1);
    for (int c = threadIdx.x; c <= g.ncell; c += RC_THREADS) bs[c] = start[c];
#pragma unroll
    for (int u = 0; u < NA; ++u) {
        const int a = threadIdx.x + u * RC_THREADS;
        if (a < N) tm[start[bsl[u] & 4095] + (bsl[u] >> 12)] = a;
    }
    __syncthreads();                                                    // (global writes of this workgroup: visible)
    // final slot = rank by atom index among the bin mates
#pragma unroll
    for (int u = 0; u < NA; ++u) {
        const int a = threadIdx.x + u * RC_THREADS;
        if (a < N) {
            const int bin = bsl[u] & 4095, s0 = start[bin], s1 = start[bin + 1];
            int rank = 0;
            for (int l = s0; l < s1; ++l) rank += tm[l] < a;
            sp[s0 + rank] = make_float4(px[u], py[u], pz[u], __int_as_float(a));
        }
    }
}

// Four atoms per wave, one DPP row (16 lanes) each: lane s of a row takes entries s, s + 16, ... of a range.  Per
// atom: the 9 stencil columns, each a contiguous range of the sorted array (the column's z-bins bz-1 .. bz+1), and at
// the z faces a second round over the single bin reached through the face.  A wave per atom spent its time waiting: own
// position -> range bounds -> candidates are three dependent round trips with four waves per SIMD to hide them (the
// 94 KB histogram allows one workgroup per CU).  With four atoms in flight per wave, the bounds of a round requested
// together and the next column's entries (four 16-entry chunks) requested while the current column is evaluated, the
// sweep is bound by its arithmetic instead.  fn(pj) runs on the lanes that hold a candidate of their row's atom.
__device__ __forceinline__ int wrap_bin(int b, int nb) { return b < 0 ? b + nb : (b >= nb ? b - nb : b); }

// HALF: only the entries ABOVE the atom's own slot in the sorted order (bin, atom index) that lie in its stencil: the
// rest of its own bin and the stencil bins with a larger linear index.  The stencil relation is symmetric, so every pair
// is visited from exactly one of its ends, no index test needed; half the columns of an interior atom drop out.
template <bool HALF, class Fn>
__device__ __forceinline__ void row_candidates(const float4* __restrict__ sp, const int32_t* __restrict__ bs,
                                               const MdgCell& cell, const CellGrid& g, const float4 pi, int slot_i,
                                               bool valid, int s, Fn&& fn) {
    const int bx = bin_coord_c(pi.x, cell.inv[0], g.nb[0]);
    const int by = bin_coord_c(pi.y, cell.inv[4], g.nb[1]);
    const int bz = bin_coord_c(pi.z, cell.inv[8], g.nb[2]);
    const int nbz = g.nb[2], zw = g.zw;
    // the z bins reached through a face: [wlo, whi] (none: wlo < 0) -- above the window's bins when it leaves through z = 0
    const int wlo = bz - zw < 0 ? nbz + bz - zw : (bz + zw > nbz - 1 ? 0 : -1);
    const int whi = bz - zw < 0 ? nbz - 1 : bz + zw - nbz;
    int col[9];
#pragma unroll
    for (int c = 0; c < 9; ++c)
        col[c] = (wrap_bin(bx + c / 3 - 1, g.nb[0]) * g.nb[1] + wrap_bin(by + c % 3 - 1, g.nb[1])) * nbz;
#pragma unroll 1
    for (int part = 0; part < 2; ++part) {
        const bool on = valid && (part == 0 || wlo >= 0);                     // (rows without an atom / a face: empty ranges)
        if (part == 1 && !__any(on)) break;
        const int zlo = part ? max(wlo, 0) : max(bz - zw, 0), zhi = part ? max(whi, 0) : min(bz + zw, nbz - 1);
        int a0[9], a1[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            a0[c] = bs[col[c] + zlo]; a1[c] = bs[col[c] + zhi + 1];
        }
        if (HALF && part == 0) a0[4] = slot_i + 1;                           // own column: from the next slot upwards
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            a0[c] += s;
            bool use = on;
            if (HALF) use = use && (c == 4 ? (part == 0 || wlo > bz) : col[c] > col[4]);
            if (!use) a1[c] = 0;
        }
        float4 P[4], Q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int a = a0[0] + 16 * k; P[k] = sp[a < a1[0] ? a : 0]; }
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            if (c + 1 < 9) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { const int a = a0[c + 1] + 16 * k; Q[k] = sp[a < a1[c + 1] ? a : 0]; }
            }
            if (__any(a0[c] < a1[c])) {                                         // (a column no row of the wave uses: skipped)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (a0[c] + 16 * k < a1[c]) fn(P[k]);
                for (int a = a0[c] + 64; a < a1[c]; a += 16) fn(sp[a]);        // (ranges beyond 64 entries: dense bins)
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) P[k] = Q[k];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// (slots are 32-bit: the entry points reject batches of 2^31 atoms or more)
__global__ __launch_bounds__(RC_THREADS) void rdf_cell_fwd_kernel(const float4* __restrict__ spos,
                                                                  const int32_t* __restrict__ bstart, int N, int total,
                                                                  MdgCell cell, CellGrid g, const float* __restrict__ mu,
                                                                  float reach, float inv_h, int nfine, float cutoff,
                                                                  uint32_t* __restrict__ ghist) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    const float tlo = -(mu[0] - reach) * inv_h;                        // fine bin of distance d: d inv_h + tlo
    for (int m = threadIdx.x; m < nfine; m += blockDim.x) hist[m] = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // (pairs beyond the list cutoff are not counted, as over a neighbour list: the stencil only reaches that far)
    const uint32_t fmax_bits = __float_as_uint(fminf((float)nfine, fmaf(cutoff, inv_h, tlo)));
    for (int t0 = (blockIdx.x * nw + wid) * 4; t0 < total; t0 += gridDim.x * nw * 4) {
        const int t = t0 + (lane >> 4);
        const bool valid = t < total;
        const int tc_ = valid ? t : total - 1, f = tc_ / N;
        const float4 pi = spos[tc_];                                   // (the row's atom: slot t - f N of the sorted frame)
        row_candidates<true>(spos + (size_t)f * N, bstart + (size_t)f * (g.ncell + 1), cell, g, pi, tc_ - f * N, valid, lane & 15,
                             [&](const float4 pj) {                       // (every pair once: the half stencil)
            float dx = pj.x - pi.x, dy = pj.y - pi.y, dz = pj.z - pi.z;
            min_image<true>(cell, dx, dy, dz);
            const float tt = fmaf(__builtin_amdgcn_sqrtf(norm2_ref(dx, dy, dz)), inv_h, tlo);
            // 0 <= tt < nfine in one unsigned compare (negative and NaN bit patterns are above every positive float)
            if (__float_as_uint(tt) < fmax_bits) atomicAdd(&hist[(int)tt], 1u);
        });
    }
    __syncthreads();
    for (int m = threadIdx.x; m < nfine; m += blockDim.x) {
        const uint32_t v = hist[m];
        if (v) atomicAdd(&ghist[m], v);
    }
}

__global__ __launch_bounds__(256) void rdf_cell_bwd_kernel(const float4* __restrict__ spos, const int32_t* __restrict__ bstart,
                                                           int N, int total, MdgCell cell, CellGrid g, MdgPairTerm term,
                                                           const float* __restrict__ theta, float* __restrict__ grad) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6, s = lane & 15;
    const TermConst tc = term_prepare(term, theta);
    for (int t0 = (blockIdx.x * nw + wid) * 4; t0 < total; t0 += gridDim.x * nw * 4) {
        const int t = t0 + (lane >> 4);
        const bool valid = t < total;
        const int tc_ = valid ? t : total - 1, f = tc_ / N;
        const float4 pi = spos[tc_];
        const int idx_i = __float_as_int(pi.w);
        float gx = 0.f, gy = 0.f, gz = 0.f;
        row_candidates<false>(spos + (size_t)f * N, bstart + (size_t)f * (g.ncell + 1), cell, g, pi, tc_ - f * N, valid, s,
                              [&](const float4 pj) {
            if (__float_as_int(pj.w) == idx_i) return;
            float dx = pj.x - pi.x, dy = pj.y - pi.y, dz = pj.z - pi.z;             // D = x_j - x_i
            min_image<true>(cell, dx, dy, dz);
            const float d2 = norm2_ref(dx, dy, dz);
            if (!(d2 < tc.rc2) || d2 == 0.f) return;
            PairOut o;
            float r, ir;
            pair_eval<1, MDG_PAIR_TABLE>(tc, d2, r, ir, o);
            const float c1 = -o.du * ir;                                               // dphi/dx_i = phi' (x_i - x_j) / d
            gx = fmaf(c1, dx, gx); gy = fmaf(c1, dy, gy); gz = fmaf(c1, dz, gz);
        });
        gx = row16_sum(gx); gy = row16_sum(gy); gz = row16_sum(gz);                  // (fixed order: reproducible)
        if (valid && s < 3) grad[((size_t)f * N + idx_i) * 3 + s] = s == 0 ? gx : (s == 1 ? gy : gz);
    }
}


// ---------------------------------------------------------------------------------------------
// column tiles (see the header): the staged neighbourhood of one (bx, by) column of one frame
constexpr int RC_MAX_NB = RC_MAX_NBZ;         // bins along z at most (x, y: 16; RC_MAX_CELLS in all)
constexpr int RC_TILE_MAX = 4096;            // staged atoms at most (64 KB of float4)

struct TileMeta {
    int cbase[10];                           // first staged slot of stencil column c (c = 3 (dx + 1) + (dy + 1)); [9] = staged atoms
    int gstart[9];                           // ... its first slot in the frame's sorted array
    int zst[9][RC_MAX_NB + 1];               // staged slot of the first atom of z-bin z of column c
    unsigned above;                          // bit c: column c's linear index is above the own column's (half stencil)
    int fits;
};

// header + staging by the whole workgroup (blockDim threads).  Ends with a barrier.
__device__ __forceinline__ void tile_stage(const float4* __restrict__ sp, const int32_t* __restrict__ bs, const CellGrid& g,
                                           int cx, int cy, int cap, TileMeta& M, float4* tile) {
    const int nbz = g.nb[2];
    __syncthreads();                                                   // (the previous tile's rows are through)
    if (threadIdx.x < 9) {
        const int c = threadIdx.x;
        const int col = (wrap_bin(cx + c / 3 - 1, g.nb[0]) * g.nb[1] + wrap_bin(cy + c % 3 - 1, g.nb[1])) * nbz;
        M.gstart[c] = bs[col];
        M.cbase[c + 1] = bs[col + nbz] - bs[col];                       // (length; prefix below)
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        unsigned above = 0u;
        const int own = (cx * g.nb[1] + cy) * nbz;
        for (int c = 0; c < 9; ++c) {
            const int len = M.cbase[c + 1];
            M.cbase[c] = run;
            run += len;
            const int col = (wrap_bin(cx + c / 3 - 1, g.nb[0]) * g.nb[1] + wrap_bin(cy + c % 3 - 1, g.nb[1])) * nbz;
            if (col > own) above |= 1u << c;
        }
        M.cbase[9] = run;
        M.above = above;
        M.fits = run <= cap;
    }
    __syncthreads();
    if (!M.fits) return;
    for (int t = threadIdx.x; t < 9 * (nbz + 1); t += blockDim.x) {
        const int c = t / (nbz + 1), z = t - c * (nbz + 1);
        const int col = (wrap_bin(cx + c / 3 - 1, g.nb[0]) * g.nb[1] + wrap_bin(cy + c % 3 - 1, g.nb[1])) * nbz;
        M.zst[c][z] = M.cbase[c] + (bs[col + z] - M.gstart[c]);
    }
    const int total = M.cbase[9];
    for (int t = threadIdx.x; t < total; t += blockDim.x) {
        int c = 0;
#pragma unroll
        for (int k = 1; k < 9; ++k) c += t >= M.cbase[k];
        tile[t] = sp[M.gstart[c] + (t - M.cbase[c])];
    }
    __syncthreads();
}

// The row sweep of row_candidates over the staged tile: the same candidates in the same per-lane order (part, column,
// entry s + 16 k), read from LDS.  slot_i = the row atom's staged slot (inside column 4).
template <bool HALF, class Fn>
__device__ __forceinline__ void tile_row_candidates(const float4* tile, const TileMeta& M, const MdgCell& cell, const CellGrid& g,
                                                    const float4 pi, int slot_i, bool valid, int s, Fn&& fn) {
    const int nbz = g.nb[2], zw = g.zw;
    const int bz = bin_coord_c(pi.z, cell.inv[8], nbz);
    const int wlo = bz - zw < 0 ? nbz + bz - zw : (bz + zw > nbz - 1 ? 0 : -1);      // (see row_candidates)
    const int whi = bz - zw < 0 ? nbz - 1 : bz + zw - nbz;
#pragma unroll 1
    for (int part = 0; part < 2; ++part) {
        const bool on = valid && (part == 0 || wlo >= 0);
        if (part == 1 && !__any(on)) break;
        const int zlo = part ? max(wlo, 0) : max(bz - zw, 0), zhi = part ? max(whi, 0) : min(bz + zw, nbz - 1);
#pragma unroll 1
        for (int c = 0; c < 9; ++c) {
            int a0 = M.zst[c][zlo], a1 = M.zst[c][zhi + 1];
            if (HALF && part == 0 && c == 4) a0 = slot_i + 1;
            bool use = on;
            if (HALF) use = use && (c == 4 ? (part == 0 || wlo > bz) : ((M.above >> c) & 1u) != 0u);
            if (!use) a1 = 0;
            for (int a = a0 + s; a < a1; a += 16) fn(tile[a]);
        }
    }
}

__global__ __launch_bounds__(RC_THREADS) void rdf_cell_fwd_tile_kernel(const float4* __restrict__ spos,
                                                                       const int32_t* __restrict__ bstart, int N, int n_frames,
                                                                       MdgCell cell, CellGrid g, const float* __restrict__ mu,
                                                                       float reach, float inv_h, int nfine, int nfine_pad,
                                                                       float cutoff, int cap, uint32_t* __restrict__ ghist) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    float4* tile = reinterpret_cast<float4*>(hist + nfine_pad);
    __shared__ TileMeta M;
    const float tlo = -(mu[0] - reach) * inv_h;
    for (int m = threadIdx.x; m < nfine; m += blockDim.x) hist[m] = 0u;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6, s = lane & 15;
    const uint32_t fmax_bits = __float_as_uint(fminf((float)nfine, fmaf(cutoff, inv_h, tlo)));
    const int ncol = g.nb[0] * g.nb[1];
    const long long ntiles = (long long)n_frames * ncol;
    auto count = [&](const float4 pi, const float4 pj) {
        float dx = pj.x - pi.x, dy = pj.y - pi.y, dz = pj.z - pi.z;
        min_image<true>(cell, dx, dy, dz);
        const float tt = fmaf(__builtin_amdgcn_sqrtf(norm2_ref(dx, dy, dz)), inv_h, tlo);
        if (__float_as_uint(tt) < fmax_bits) atomicAdd(&hist[(int)tt], 1u);
    };
    for (long long tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int f = (int)(tl / ncol), colid = (int)(tl - (long long)f * ncol);
        const float4* sp = spos + (size_t)f * N;
        const int32_t* bs = bstart + (size_t)f * (g.ncell + 1);
        tile_stage(sp, bs, g, colid / g.nb[1], colid % g.nb[1], cap, M, tile);
        const int own0 = M.cbase[4], nown = M.cbase[5] - M.cbase[4], g0 = M.gstart[4];
        for (int r0 = wid * 4; r0 < nown; r0 += nw * 4) {
            const int r = r0 + (lane >> 4);
            const bool valid = r < nown;
            const int rc_ = valid ? r : nown - 1;
            if (M.fits) {
                const float4 pi = tile[own0 + rc_];
                tile_row_candidates<true>(tile, M, cell, g, pi, own0 + rc_, valid, s, [&](const float4 pj) { count(pi, pj); });
            } else {
                const float4 pi = sp[g0 + rc_];
                row_candidates<true>(sp, bs, cell, g, pi, g0 + rc_, valid, s, [&](const float4 pj) { count(pi, pj); });
            }
        }
    }
    __syncthreads();
    for (int m = threadIdx.x; m < nfine; m += blockDim.x) {
        const uint32_t v = hist[m];
        if (v) atomicAdd(&ghist[m], v);
    }
}

// (Also measured and dropped in round 5: bins of HALF the cutoff with a 5 x 5 x 5 stencil -- 42 % fewer candidates for the same
//  pairs -- as tiles of 2 x 2 columns staging 6 x 6: histogram identical, forward 3.7 ms against 1.5 ms, backward 3.4 against
//  2.7 for 704 frames of 4 096 atoms.  Twenty-five z-ranges of ~12 atoms per row instead of nine of ~57: the per-column
//  bookkeeping and the 75 %-full 16-lane passes cost more than the candidates saved.)
// (Measured and dropped in round 5: a TEST / EVALUATE split of this sweep -- rows only test their candidates and append the
//  accepted ones, one in eight, to an LDS queue, the table force then runs with full lanes.  3.29 ms against 2.66 ms for 704
//  frames of 4 096 atoms: the ballot / queue-write chain serialises every 16-candidate step behind its LDS read, and the
//  straight loop below, which the compiler unrolls and overlaps freely, hides that latency better than the split saves work.)
__global__ __launch_bounds__(256) void rdf_cell_bwd_tile_kernel(const float4* __restrict__ spos, const int32_t* __restrict__ bstart,
                                                                int N, int n_frames, MdgCell cell, CellGrid g, MdgPairTerm term,
                                                                const float* __restrict__ theta, int cap, float* __restrict__ grad) {
    extern __shared__ __attribute__((aligned(16))) float4 tile_b[];
    __shared__ TileMeta M;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6, s = lane & 15;
    const TermConst tc = term_prepare(term, theta);
    const int ncol = g.nb[0] * g.nb[1];
    const long long ntiles = (long long)n_frames * ncol;
    for (long long tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int f = (int)(tl / ncol), colid = (int)(tl - (long long)f * ncol);
        const float4* sp = spos + (size_t)f * N;
        const int32_t* bs = bstart + (size_t)f * (g.ncell + 1);
        tile_stage(sp, bs, g, colid / g.nb[1], colid % g.nb[1], cap, M, tile_b);
        const int own0 = M.cbase[4], nown = M.cbase[5] - M.cbase[4], g0 = M.gstart[4];
        for (int r0 = wid * 4; r0 < nown; r0 += nw * 4) {
            const int r = r0 + (lane >> 4);
            const bool valid = r < nown;
            const int rc_ = valid ? r : nown - 1;
            const float4 pi = M.fits ? tile_b[own0 + rc_] : sp[g0 + rc_];
            const int idx_i = __float_as_int(pi.w);
            float gx = 0.f, gy = 0.f, gz = 0.f;
            auto force = [&](const float4 pj) {
                if (__float_as_int(pj.w) == idx_i) return;
                float dx = pj.x - pi.x, dy = pj.y - pi.y, dz = pj.z - pi.z;             // D = x_j - x_i
                min_image<true>(cell, dx, dy, dz);
                const float d2 = norm2_ref(dx, dy, dz);
                if (!(d2 < tc.rc2) || d2 == 0.f) return;
                PairOut o;
                float rr, ir;
                pair_eval<1, MDG_PAIR_TABLE>(tc, d2, rr, ir, o);
                const float c1 = -o.du * ir;
                gx = fmaf(c1, dx, gx); gy = fmaf(c1, dy, gy); gz = fmaf(c1, dz, gz);
            };
            if (M.fits) tile_row_candidates<false>(tile_b, M, cell, g, pi, own0 + rc_, valid, s, force);
            else row_candidates<false>(sp, bs, cell, g, pi, g0 + rc_, valid, s, force);
            gx = row16_sum(gx); gy = row16_sum(gy); gz = row16_sum(gz);                  // (fixed order: reproducible)
            if (valid && s < 3) grad[((size_t)f * N + idx_i) * 3 + s] = s == 0 ? gx : (s == 1 ? gy : gz);
        }
    }
}

constexpr size_t RC_LDS_BYTES = 160 * 1024;   // LDS of a gfx950 CU (one 1 024-thread forward workgroup per CU)

// MDG_RDF_CELL_TILES=0 keeps the row sweeps from L2 (A/B measurements, tests of the two against each other)
bool tiles_enabled() {
    const char* e = getenv("MDG_RDF_CELL_TILES");      // (read per call: a test flips it between two calls)
    return !(e && e[0] == '0');
}

// staged capacity of a launch: 1.5 x the nine columns of a uniform frame (+ slack), at most RC_TILE_MAX and what `lds_left`
// bytes hold; tiles beyond it take the row sweep from L2
int tile_capacity(int n_atoms, const CellGrid& g, size_t lds_left) {
    long long est = (long long)(9.0 * 1.5 * n_atoms / (double)(g.nb[0] * g.nb[1])) + 64;
    if (est > n_atoms) est = n_atoms;
    if (est > RC_TILE_MAX) est = RC_TILE_MAX;
    const long long room = (long long)(lds_left / sizeof(float4));
    if (est > room) est = room;
    return est < 64 ? 0 : (int)est;
}

struct Scratch { int32_t* bstart; int32_t* tmp; float4* spos; };

Scratch carve(int32_t* scratch, int F, int N, const CellGrid& g) {
    Scratch s;
    s.spos = reinterpret_cast<float4*>(scratch);                                      // (16-byte aligned: first)
    s.tmp = scratch + (size_t)4 * F * N;
    s.bstart = s.tmp + (size_t)F * N;
    return s;
}

bool grid_ok(int n_atoms, const MdgCell* cell, float cutoff, CellGrid* g) {
    if (!cell || !cell->diag || n_atoms <= 0 || n_atoms > RC_MAX_ATOMS || !(cutoff > 0.f)) return false;
    *g = make_grid(*cell, cutoff);
    return g->nb[0] >= 3 && g->nb[1] >= 3 && g->nb[2] >= 3 && g->ncell <= RC_MAX_CELLS;
}

}  // namespace

extern "C" int mdg_rdf_cell_supported(int n_atoms, const MdgCell* cell, float cutoff) {
    CellGrid g;
    return grid_ok(n_atoms, cell, cutoff, &g) ? 1 : 0;
}

extern "C" int64_t mdg_rdf_cell_scratch(int n_frames, int n_atoms, const MdgCell* cell, float cutoff) {
    CellGrid g;
    if (n_frames <= 0 || !grid_ok(n_atoms, cell, cutoff, &g)) return -1;
    return (int64_t)5 * n_frames * n_atoms + (int64_t)n_frames * (g.ncell + 1);
}

extern "C" int mdg_rdf_fwd_cell(const float* xyz, int n_frames, int n_atoms, const MdgCell* cell, float cutoff,
                                const float* mu, float spacing, float coeff, int nbins, float* raw, int32_t* scratch,
                                void* stream) {
    MDG_CHECK_ARG(xyz && cell && mu && raw && scratch && n_frames > 0 && nbins > 0, "rdf_fwd_cell: bad arguments");
    CellGrid g;
    MDG_CHECK_ARG(grid_ok(n_atoms, cell, cutoff, &g), "rdf_fwd_cell: needs an orthorhombic cell of >= 3 cutoffs per side and "
                  "at most %d atoms (see mdg_rdf_cell_supported)", RC_MAX_ATOMS);
    MDG_CHECK_ARG((reinterpret_cast<uintptr_t>(scratch) & 15) == 0, "rdf_fwd_cell: scratch must be 16-byte aligned");
    const RdfFinePlan P = mdg_rdf_fine_plan(spacing, coeff, nbins);
    MDG_CHECK_ARG(P.nfine > 0, "rdf_fwd_cell: the fine grid for these centres does not fit (see mdg_rdf_ell_supported)");
    hipStream_t st = (hipStream_t)stream;
    const Scratch S = carve(scratch, n_frames, n_atoms, g);
    if (n_atoms <= 4 * RC_THREADS)
        hipLaunchKernelGGL((rdf_cell_bin_kernel<4>), dim3(n_frames), dim3(RC_THREADS), 0, st, xyz, n_atoms, *cell, g, S.bstart,
                           S.tmp, S.spos);
    else if (n_atoms <= 16 * RC_THREADS)
        hipLaunchKernelGGL((rdf_cell_bin_kernel<16>), dim3(n_frames), dim3(RC_THREADS), 0, st, xyz, n_atoms, *cell, g, S.bstart,
                           S.tmp, S.spos);
    else
        hipLaunchKernelGGL((rdf_cell_bin_kernel<32>), dim3(n_frames), dim3(RC_THREADS), 0, st, xyz, n_atoms, *cell, g, S.bstart,
                           S.tmp, S.spos);
    uint32_t* ghist = nullptr;
    MDG_HIP(hipMallocAsync((void**)&ghist, sizeof(uint32_t) * (size_t)P.nfine, st));
    MDG_HIP(hipMemsetAsync(ghist, 0, sizeof(uint32_t) * (size_t)P.nfine, st));
    MDG_CHECK_ARG((long long)n_frames * n_atoms < (1ll << 31), "rdf_fwd_cell: fewer than 2^31 atoms per call");
    const int total = n_frames * n_atoms;
    int grid = (total + 63) / 64;
    if (grid > 512) grid = 512;                                      // (persistent: one histogram flush per workgroup)
    const int nfine_pad = ((int)P.nfine + 3) & ~3;
    const size_t hist_bytes = sizeof(uint32_t) * (size_t)nfine_pad;
    const int cap = (tiles_enabled() && hist_bytes + 4096 < RC_LDS_BYTES) ? tile_capacity(n_atoms, g, RC_LDS_BYTES - 4096 - hist_bytes) : 0;
    if (cap > 0) {
        const long long ntiles = (long long)n_frames * g.nb[0] * g.nb[1];
        const int gt = ntiles < 512 ? (int)ntiles : 512;
        hipLaunchKernelGGL(rdf_cell_fwd_tile_kernel, dim3((unsigned)gt), dim3(RC_THREADS), hist_bytes + sizeof(float4) * (size_t)cap,
                           st, S.spos, S.bstart, n_atoms, n_frames, *cell, g, mu, P.reach, 1.0f / P.h, (int)P.nfine, nfine_pad,
                           cutoff, cap, ghist);
    } else
    hipLaunchKernelGGL(rdf_cell_fwd_kernel, dim3((unsigned)grid), dim3(RC_THREADS), sizeof(uint32_t) * (size_t)P.nfine, st,
                       S.spos, S.bstart, n_atoms, total, *cell, g, mu, P.reach, 1.0f / P.h, (int)P.nfine, cutoff, ghist);
    const int rc = mdg_rdf_fine_finish(ghist, P, mu, nbins, raw, st);
    (void)hipFreeAsync(ghist, st);
    if (rc) return rc;
    MDG_CHECK_LAUNCH("rdf_fwd_cell");
    return MDG_OK;
}

extern "C" int mdg_rdf_bwd_cell(int n_frames, int n_atoms, const MdgCell* cell, float cutoff, const MdgPairTerm* term,
                                const float* theta, const int32_t* scratch, float* g_xyz, void* stream) {
    MDG_CHECK_ARG(cell && term && theta && scratch && g_xyz && n_frames > 0, "rdf_bwd_cell: bad arguments");
    MDG_CHECK_ARG(term->kind == MDG_PAIR_TABLE && !term->mask, "rdf_bwd_cell: the term must be an unmasked MDG_PAIR_TABLE");
    CellGrid g;
    MDG_CHECK_ARG(grid_ok(n_atoms, cell, cutoff, &g), "rdf_bwd_cell: unsupported cell / size (see mdg_rdf_cell_supported)");
    hipStream_t st = (hipStream_t)stream;
    const Scratch S = carve(const_cast<int32_t*>(scratch), n_frames, n_atoms, g);
    MDG_CHECK_ARG((long long)n_frames * n_atoms < (1ll << 31), "rdf_bwd_cell: fewer than 2^31 atoms per call");
    const int total = n_frames * n_atoms;
    int grid = (total + 15) / 16;
    if (grid > 4096) grid = 4096;
    const int cap = tiles_enabled() ? tile_capacity(n_atoms, g, sizeof(float4) * (size_t)RC_TILE_MAX) : 0;
    if (cap > 0) {
        const long long ntiles = (long long)n_frames * g.nb[0] * g.nb[1];
        const int gt = ntiles < (1 << 20) ? (int)ntiles : (1 << 20);
        hipLaunchKernelGGL(rdf_cell_bwd_tile_kernel, dim3((unsigned)gt), dim3(256), sizeof(float4) * (size_t)cap, st, S.spos, S.bstart,
                           n_atoms, n_frames, *cell, g, *term, theta, cap, g_xyz);
    } else
    hipLaunchKernelGGL(rdf_cell_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, st, S.spos, S.bstart, n_atoms, total, *cell, g,
                       *term, theta, g_xyz);
    MDG_CHECK_LAUNCH("rdf_bwd_cell");
    return MDG_OK;
}
