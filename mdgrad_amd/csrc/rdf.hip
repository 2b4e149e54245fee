// K8: soft-histogram radial distribution function and its gradient.
// Replaces rdf.forward (torchmd/observable.py:62-76: generate_nbr_list with cutoff end+0.5
// over all frames -> GaussianSmearing nff/nn/layers.py:14-31 -> sum) and the autograd
// backward through it.  The [pairs, bins] exp matrix of the reference is never
// materialised.
//
// forward : few frames / arbitrary centres -- persistent blocks stride over (frame, candidate chunk) work
//           items, compact the accepted i<j distances into LDS in a fixed order, thread k owns bin k (or
//           an 8-bin block with the Gaussian recurrence) and sweeps them (rdf_fwd_kernel,
//           rdf_fwd_block8_kernel).  Many frames, equally spaced centres (the replica-batched training
//           shape) -- one wave per frame, one lane per pair, lane-private LDS histogram columns
//           (rdf_fwd_half_kernel / rdf_fwd_lane_kernel).  Per-block / per-wave partial histograms are added
//           by a second kernel in a fixed order => no atomics, reproducible.  exp is v_exp_f32.
// backward: each accepted pair contributes dL/dd = sum_k g_raw[k] * 2 coeff (d - mu_k) e_k along +/- its
//           unit separation vector.  Many frames: one wave per frame, dL/dd from a fine table, cyclic pair
//           order with one end's gradient in registers (rdf_bwd_fine_kernel) or the bin recurrence in
//           round-robin-tournament order (rdf_bwd_kernel); few frames: (frame, atom) gather
//           (rdf_bwd_atom_kernel).  LDS accumulation without atomics.
#include "common.hpp"
#include <type_traits>

namespace {

constexpr int RDF_BLOCK = 512;
constexpr int RDF_MAX_BLOCKS = 2048;     // persistent blocks: each strides over (frame, chunk) work items
constexpr int RDF_CHUNK = 8192;          // candidate pairs per work item
constexpr float LOG2E = 1.4426950408889634f;

// flat index c in [0, N(N-1)/2) -> (i, j), i < j, row-major (the order torch.nonzero yields)
__device__ __forceinline__ void pair_from_flat(long long c, int N, int& i, int& j) {
    const double b = 2.0 * N - 1.0;
    int ii = (int)((b - sqrt(b * b - 8.0 * (double)c)) * 0.5);
    while ((long long)ii * (2LL * N - ii - 1) / 2 > c) --ii;
    while ((long long)(ii + 1) * (2LL * N - ii - 2) / 2 <= c) ++ii;
    i = ii;
    j = (int)(c - (long long)ii * (2LL * N - ii - 1) / 2) + ii + 1;
}

// exp(coeff (d - mu)^2) = exp2(-(s d - s mu)^2) with s = sqrt(-coeff log2 e): distances and
// the sweep costs sub, mul, mul, v_exp_f32, add per (pair, bin).
//
// Thread layout: nbins <= RDF_BLOCK.  G = RDF_BLOCK / nbins thread groups; thread (g, k) owns
// bin k and sweeps the compacted distances 4g..4g+3, 4(g+G).., ... (ds_read_b128); the groups
// are combined in group order at the end.  Bins beyond RDF_BLOCK are handled by the slow path.
template <bool DIAG>
__global__ __launch_bounds__(RDF_BLOCK) void rdf_fwd_kernel(
    const float* __restrict__ xyz, int nF, int N, MdgCell cell, float rc2, const uint8_t* __restrict__ mask,
    const float* __restrict__ mu, float coeff, int nbins, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float dist[RDF_BLOCK + 4];
    __shared__ int wcnt[RDF_BLOCK / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    constexpr int nw = RDF_BLOCK / 64;
    const float sc = sqrtf(-coeff * LOG2E);
    const int G = RDF_BLOCK / nbins;
    const int grp = threadIdx.x / nbins;
    const bool active = grp < G;
    const int k = threadIdx.x - grp * nbins;
    const float ms = active ? mu[k] : 0.f;
    float acc = 0.f;
    const long long npair = (long long)N * (N - 1) / 2;
    const int chunks = (int)((npair + RDF_CHUNK - 1) / RDF_CHUNK);
    const long long items = (long long)nF * chunks;
    for (long long it = blockIdx.x; it < items; it += gridDim.x) {
        const int fr = (int)(it / chunks), ch = (int)(it % chunks);
        const float* pos = xyz + (size_t)fr * N * 3;
        const long long c_end = min(npair, (long long)(ch + 1) * RDF_CHUNK);
        long long c = (long long)ch * RDF_CHUNK + threadIdx.x;
        int i = 0, j = 0;
        if (c < c_end) pair_from_flat(c, N, i, j);
        for (long long cb = (long long)ch * RDF_CHUNK; cb < c_end; cb += RDF_BLOCK) {
            float d = -1.f;
            if (c < c_end) {
                float dx = pos[3 * j] - pos[3 * i], dy = pos[3 * j + 1] - pos[3 * i + 1],
                      dz = pos[3 * j + 2] - pos[3 * i + 2];
                min_image<DIAG>(cell, dx, dy, dz);
                const float d2 = norm2_ref(dx, dy, dz);
                bool ok = (d2 < rc2) && (d2 != 0.f);
                if (ok && mask) ok = mask[(size_t)i * N + j] != 0;
                if (ok) d = sqrtf(d2);
                // advance to the candidate RDF_BLOCK further on
                c += RDF_BLOCK;
                j += RDF_BLOCK;
                while (j >= N && i < N - 1) { ++i; j = j - N + i + 1; }
            }
            const unsigned long long b = __ballot(d >= 0.f);
            __syncthreads();                                   // previous sweep done with dist[]
            if (lane == 0) wcnt[wid] = __popcll(b);
            __syncthreads();
            int base = 0, total = 0;
#pragma unroll
            for (int w = 0; w < nw; ++w) { if (w < wid) base += wcnt[w]; total += wcnt[w]; }
            if (d >= 0.f) dist[base + __popcll(b & ((1ull << lane) - 1ull))] = d;
            if (threadIdx.x < 4) dist[total + threadIdx.x] = 3.0e18f;   // pad: exp2(-x^2) == 0
            __syncthreads();
            if (active) {
                for (int p = 4 * grp; p < total; p += 4 * G) {
                    const float4 dd = *reinterpret_cast<const float4*>(&dist[p]);
                    // (d - mu) first, then scale: no cancellation error on the scaled values
                    const float x0 = (dd.x - ms) * sc, x1 = (dd.y - ms) * sc, x2 = (dd.z - ms) * sc,
                                x3 = (dd.w - ms) * sc;
                    acc += __builtin_amdgcn_exp2f(-x0 * x0);
                    acc += __builtin_amdgcn_exp2f(-x1 * x1);
                    acc += __builtin_amdgcn_exp2f(-x2 * x2);
                    acc += __builtin_amdgcn_exp2f(-x3 * x3);
                }
            }
        }
    }
    __syncthreads();
    if (active) dist[threadIdx.x] = acc;                        // [g][k], G * nbins <= RDF_BLOCK
    __syncthreads();
    if (threadIdx.x < nbins) {
        float s = 0.f;
        for (int g = 0; g < G; ++g) s += dist[g * nbins + threadIdx.x];
        partial[(size_t)blockIdx.x * nbins + threadIdx.x] = s;
    }
}

// Variant for equally spaced centres with Ds = s * spacing <= 1 (the default width == spacing gives
// Ds = 0.85): a thread owns a block of 8 consecutive bins and gets their 8 Gaussians of one distance
// from ONE centre evaluation and a two-term recurrence outwards from the block's middle bin
//     e_{j+1} = e_j rho_j , rho_{j+1} = rho_j exp2(-2 Ds^2) , rho = exp2(+-2 Ds x - Ds^2)
// i.e. 3 v_exp_f32 + 14 multiplies per 8 (pair, bin) values instead of 8 exps + 16 multiplies.  The
// recurrence runs at most 4 steps (relative error ~1e-6); a block whose middle bin is farther than the
// exp2 underflow reach from the distance can only miss terms below 2^-60.
constexpr int RDF_KB = 8;

template <bool DIAG>
__global__ __launch_bounds__(RDF_BLOCK) void rdf_fwd_block8_kernel(
    const float* __restrict__ xyz, int nF, int N, MdgCell cell, float rc2, const uint8_t* __restrict__ mask,
    const float* __restrict__ mu, float coeff, int nbins, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* dist = sm;                                   // [RDF_BLOCK]
    int* wcnt = (int*)(sm + RDF_BLOCK);                 // [nw]
    float* comb = sm + RDF_BLOCK + 16;                  // [G][nblk*8] final combine
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    constexpr int nw = RDF_BLOCK / 64;
    const float sc = sqrtf(-coeff * LOG2E);
    const int nblk = (nbins + RDF_KB - 1) / RDF_KB;     // bin blocks
    const int G = RDF_BLOCK / nblk;                     // thread groups sharing the sweep
    const int grp = threadIdx.x / nblk;
    const bool active = grp < G;
    const int kb = threadIdx.x - grp * nblk;            // this thread's bin block
    const float mu0 = mu[0];
    const float dmu = (mu[nbins - 1] - mu0) / (float)(nbins - 1);
    const float Ds = dmu * sc;
    const float c2 = __builtin_amdgcn_exp2f(-2.f * Ds * Ds);
    const int kmid = min(nbins - 1, kb * RDF_KB + 4);
    const float mum = active ? mu[kmid] : 0.f;          // centre of the block's middle bin (exact value)
    const float off = (float)(kb * RDF_KB + 4 - kmid) * dmu;   // 0 unless the last block is partial
    float acc[RDF_KB];
#pragma unroll
    for (int j = 0; j < RDF_KB; ++j) acc[j] = 0.f;
    const long long npair = (long long)N * (N - 1) / 2;
    const int chunks = (int)((npair + RDF_CHUNK - 1) / RDF_CHUNK);
    const long long items = (long long)nF * chunks;
    for (long long it = blockIdx.x; it < items; it += gridDim.x) {
        const int fr = (int)(it / chunks), ch = (int)(it % chunks);
        const float* pos = xyz + (size_t)fr * N * 3;
        const long long c_end = min(npair, (long long)(ch + 1) * RDF_CHUNK);
        long long c = (long long)ch * RDF_CHUNK + threadIdx.x;
        int i = 0, j = 0;
        if (c < c_end) pair_from_flat(c, N, i, j);
        for (long long cb = (long long)ch * RDF_CHUNK; cb < c_end; cb += RDF_BLOCK) {
            float d = -1.f;
            if (c < c_end) {
                float dx = pos[3 * j] - pos[3 * i], dy = pos[3 * j + 1] - pos[3 * i + 1],
                      dz = pos[3 * j + 2] - pos[3 * i + 2];
                min_image<DIAG>(cell, dx, dy, dz);
                const float d2 = norm2_ref(dx, dy, dz);
                bool ok = (d2 < rc2) && (d2 != 0.f);
                if (ok && mask) ok = mask[(size_t)i * N + j] != 0;
                if (ok) d = sqrtf(d2);
                c += RDF_BLOCK;
                j += RDF_BLOCK;
                while (j >= N && i < N - 1) { ++i; j = j - N + i + 1; }
            }
            const unsigned long long b = __ballot(d >= 0.f);
            __syncthreads();                                   // previous sweep done with dist[]
            if (lane == 0) wcnt[wid] = __popcll(b);
            __syncthreads();
            int base = 0, total = 0;
#pragma unroll
            for (int w = 0; w < nw; ++w) { if (w < wid) base += wcnt[w]; total += wcnt[w]; }
            if (d >= 0.f) dist[base + __popcll(b & ((1ull << lane) - 1ull))] = d;
            __syncthreads();
            if (active) {
                for (int p = grp; p < total; p += G) {
                    const float x4 = (dist[p] - mum - off) * sc;          // x at block bin 4
                    // beyond 12 the block's nearest bin is > 12 - 4 Ds >= 8 away: all eight terms are
                    // below 2^-64 (and the ratio below would overflow against an underflowed e4)
                    if (fabsf(x4) >= 12.f) continue;
                    const float e4 = __builtin_amdgcn_exp2f(-x4 * x4);
                    float eu = e4, ed = e4;
                    float ru = __builtin_amdgcn_exp2f(2.f * Ds * x4 - Ds * Ds);     // towards larger mu
                    float rd = __builtin_amdgcn_exp2f(-2.f * Ds * x4 - Ds * Ds);    // towards smaller mu
                    acc[4] += e4;
                    eu *= ru; ru *= c2; acc[5] += eu;
                    ed *= rd; rd *= c2; acc[3] += ed;
                    eu *= ru; ru *= c2; acc[6] += eu;
                    ed *= rd; rd *= c2; acc[2] += ed;
                    eu *= ru;           acc[7] += eu;
                    ed *= rd; rd *= c2; acc[1] += ed;
                    ed *= rd;           acc[0] += ed;
                }
            }
        }
    }
    __syncthreads();
    const int nk = nblk * RDF_KB;
    if (active) {
#pragma unroll
        for (int j = 0; j < RDF_KB; ++j) comb[grp * nk + kb * RDF_KB + j] = acc[j];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nbins; k += RDF_BLOCK) {
        float s = 0.f;
        for (int g = 0; g < G; ++g) s += comb[g * nk + k];
        partial[(size_t)blockIdx.x * nbins + k] = s;
    }
}


// Equally spaced centres, many frames (the replica-batched training shape): one WAVE per frame, one LANE
// per pair.  Each lane owns a private column of a wave-private LDS histogram [rows][64], so a pair's
// Gaussians are deposited with plain (conflict-free, order-fixed) LDS adds -- no bin-owner sweep, every
// pair is looked at exactly once.  Only the 2R+1 bins around the nearest centre kc are touched: the
// centre value and two outward recurrences (as in the block-of-8 kernel); the nearest bin left out is
// (R + 1/2) Ds >= 4.65 away, where the Gaussian is below 2^-21.6 of the peak: a relative 3e-7 of a bin's count
// in the worst case, under the fp32 rounding of a sum of thousands of terms.  Real bin k lives in row
// k + 2R; rows outside [2R, 2R + nbins) are write-only padding so that no deposit needs a bounds test.
// The columns are summed in a fixed order at the end => bitwise reproducible.
//
// The pairs come from a table built once per call (it stays in L1/L2): entry = (4 i) | (4 j2) << 16 stands for
// the TWO pairs (i, j2), (i, j2 + 1) with j2 even, rows in order, j2 from (i + 1) & ~1.  No index arithmetic in
// the loop (the closed-form round-robin order this replaces spent 27 of its 95 VALU instructions per pair on
// it); the i coordinates are an LDS broadcast, the two j's one aligned ds_read_b64 per component, already in
// packed-fp32 register pairs.  Slots that are not pairs need no flags: (i, i) -- the first slot of an even
// row -- is rejected by the d2 != 0 test of topology.py:67, column N of an odd N and the columns the padding
// entries point at hold NaN, and NaN < cutoff^2 is false.
__host__ __device__ inline int rdf_row_entries(int i, int N) { return (N + 1 - ((i + 1) & ~1)) >> 1; }

__global__ __launch_bounds__(256) void rdf_pair_table_kernel(uint32_t* __restrict__ tab, int N, int n_entries,
                                                             int n_padded, uint32_t pad_entry) {
    __shared__ int off_s;
    const int i = blockIdx.x;
    if (i < N - 1) {
        if (threadIdx.x == 0) off_s = 0;
        __syncthreads();
        int part = 0;
        for (int k = threadIdx.x; k < i; k += blockDim.x) part += rdf_row_entries(k, N);
        if (part) atomicAdd(&off_s, part);
        __syncthreads();
        const int off = off_s, j0 = (i + 1) & ~1, cnt = rdf_row_entries(i, N);
        for (int e = threadIdx.x; e < cnt; e += blockDim.x)
            tab[off + e] = (uint32_t)(4 * i) | ((uint32_t)(4 * (j0 + 2 * e)) << 16);
    } else {
        for (int p_ = n_entries + threadIdx.x; p_ < n_padded; p_ += blockDim.x) tab[p_] = pad_entry;
    }
}

constexpr int RDF_TABLE_MAX_ATOMS = 4096;      // 4 j < 2^16 and a table of at most 16 MB

// Constants of the deposit: e_{+-s} = e0 r^s K_s with r = exp2(+-2 Ds x0 - Ds^2) <= 1 and K_s = c2^(s(s-1)/2),
// K_0 = K_1 = 1.  Bins are handled two adjacent rows at a time (one ds_read2st64 / ds_write2st64 and one packed
// fma per two bins): the packed power {r^s, r^(s+1)} advances by r^2.  Outward chain s = 0 (the centre) .. R,
// inward chain s = 1 .. R.
template <int R>
struct RdfLane {
    static constexpr int NPO = (R + 1) / 2, NPI = R / 2;
    static constexpr bool TAIL_O = ((R + 1) & 1) != 0, TAIL_I = (R & 1) != 0;   // one unpaired row at the far end
    float sc, mu0, inv_dmu, dmu, Ds, Ds2, Klast;
    f32x2 Ko[NPO], Ki[NPI > 0 ? NPI : 1];                       // outward {K_2m, K_2m+1}, inward {K_2m+2, K_2m+1}
    struct Rows { f32x2 vo[NPO]; f32x2 vi[NPI > 0 ? NPI : 1]; float vlo, vli; float* h; };

    __device__ __forceinline__ void init(const float* __restrict__ mu, float coeff, int nbins) {
        sc = sqrtf(-coeff * LOG2E);
        mu0 = mu[0];
        dmu = (mu[nbins - 1] - mu0) / (float)(nbins - 1);
        inv_dmu = 1.f / dmu;
        Ds = dmu * sc; Ds2 = Ds * Ds;
        const float c2 = __builtin_amdgcn_exp2f(-2.f * Ds2);
        float cp = 1.f, K = 1.f, Ks[R + 2];
        Ks[0] = 1.f; Ks[1] = 1.f;
#pragma unroll
        for (int s_ = 2; s_ <= R; ++s_) { cp *= c2; K *= cp; Ks[s_] = K; }
#pragma unroll
        for (int m = 0; m < NPO; ++m) Ko[m] = f32x2{Ks[2 * m], Ks[2 * m + 1]};
#pragma unroll
        for (int m = 0; m < NPI; ++m) Ki[m] = f32x2{Ks[2 * m + 2], Ks[2 * m + 1]};
        Klast = Ks[R];
    }
    // the 2R + 1 rows around bin kc of this lane's column (col = hist + lane)
    __device__ __forceinline__ void rows_load(float* col, int kc, Rows& w) const {
        float* h = col + (size_t)(kc + R) * 64;                  // row of bin kc - R
        w.h = h;
#pragma unroll
        for (int m_ = 0; m_ < NPO; ++m_) w.vo[m_] = f32x2{h[(R + 2 * m_) * 64], h[(R + 2 * m_ + 1) * 64]};
#pragma unroll
        for (int m_ = 0; m_ < NPI; ++m_) w.vi[m_] = f32x2{h[(R - 2 * m_ - 2) * 64], h[(R - 2 * m_ - 1) * 64]};
        w.vlo = 0.f; w.vli = 0.f;
        if (TAIL_O) w.vlo = h[2 * R * 64];
        if (TAIL_I) w.vli = h[0];
    }
    // branch-free: a rejected pair adds +0 (a bitwise no-op) around bin 0.  Plain read-add-write on the
    // lane-private column: ds_add_f32 is serialised per lane in the LDS atomic unit (measured 9x slower).
    __device__ __forceinline__ void rows_add_store(Rows& w, float d, float m, bool ok) const {
        const float x0 = (d - m) * sc;
        const float a_ = ok ? 2.f * Ds * x0 : 0.f;
        const float e0 = ok ? __builtin_amdgcn_exp2f(-x0 * x0) : 0.f;
        const float rp = __builtin_amdgcn_exp2f(a_ - Ds2), rm = __builtin_amdgcn_exp2f(-a_ - Ds2);
        const float rp2 = rp * rp, rm2 = rm * rm;
        f32x2 Po = e0 * f32x2{1.f, rp}, Pi = e0 * f32x2{rm2, rm};
#pragma unroll
        for (int m_ = 0; m_ < NPO; ++m_) {
            w.vo[m_] += Ko[m_] * Po;
            if (m_ + 1 < NPO || TAIL_O) Po *= rp2;
        }
#pragma unroll
        for (int m_ = 0; m_ < NPI; ++m_) {
            w.vi[m_] += Ki[m_] * Pi;
            if (m_ + 1 < NPI || TAIL_I) Pi *= rm2;
        }
        if (TAIL_O) w.vlo += Klast * Po.x;                       // Po = {e0 r^R, .} after NPO steps (R even)
        if (TAIL_I) w.vli += Klast * Pi.y;                       // Pi = {., e0 r^R} after NPI steps (R odd)
        float* h = w.h;
#pragma unroll
        for (int m_ = 0; m_ < NPO; ++m_) { h[(R + 2 * m_) * 64] = w.vo[m_].x; h[(R + 2 * m_ + 1) * 64] = w.vo[m_].y; }
#pragma unroll
        for (int m_ = 0; m_ < NPI; ++m_) { h[(R - 2 * m_ - 2) * 64] = w.vi[m_].x; h[(R - 2 * m_ - 1) * 64] = w.vi[m_].y; }
        if (TAIL_O) h[2 * R * 64] = w.vlo;
        if (TAIL_I) h[0] = w.vli;
    }
};

// the table entry's two pairs: squared minimum-image distances in packed fp32 (px: SoA rows of stride PXLD)
template <bool DIAG, bool NEAR>
__device__ __forceinline__ f32x2 rdf_entry_d2(const float* px, int PXLD, const MdgCell& cell, float ivx, float ivy, float ivz,
                                              uint32_t t, int& i4, int& j4) {
    i4 = (int)(t & 0xffffu); j4 = (int)(t >> 16);
    const float* pi = reinterpret_cast<const float*>(reinterpret_cast<const char*>(px) + i4);
    const float* pj = reinterpret_cast<const float*>(reinterpret_cast<const char*>(px) + j4);
    f32x2 dx = *reinterpret_cast<const f32x2*>(pj) - pi[0], dy = *reinterpret_cast<const f32x2*>(pj + PXLD) - pi[PXLD],
          dz = *reinterpret_cast<const f32x2*>(pj + 2 * PXLD) - pi[2 * PXLD];
    if constexpr (NEAR) {
        dx = min_image_diag2_near(dx, ivx, cell.h[0]);
        dy = min_image_diag2_near(dy, ivy, cell.h[4]);
        dz = min_image_diag2_near(dz, ivz, cell.h[8]);
    } else if constexpr (DIAG) {
        dx = min_image_diag2(dx, ivx, cell.h[0]);
        dy = min_image_diag2(dy, ivy, cell.h[4]);
        dz = min_image_diag2(dz, ivz, cell.h[8]);
    } else {
        float ax_ = dx.x, ay_ = dy.x, az_ = dz.x, bx_ = dx.y, by_ = dy.y, bz_ = dz.y;
        min_image<false>(cell, ax_, ay_, az_);
        min_image<false>(cell, bx_, by_, bz_);
        dx = f32x2{ax_, bx_}; dy = f32x2{ay_, by_}; dz = f32x2{az_, bz_};
    }
    return norm2_ref2(dx, dy, dz);
}

// a frame's coordinates into SoA LDS rows; true when every atom is within [-0.24, 1.24] cell lengths
// (trajectories are wrapped at every epoch, md.py:66): the image shift is then a plain rint, see
// min_image_diag2_near
template <bool DIAG>
__device__ __forceinline__ bool rdf_load_frame(float* px, int PXLD, const float* __restrict__ pos, int N, int lane,
                                               float ivx, float ivy, float ivz) {
    bool out = false;
    for (int e = lane; e < 3 * N; e += 64) {
        const int c = e % 3;
        const float v = pos[e];
        px[c * PXLD + e / 3] = v;
        if (DIAG) {
            const float s_ = v * (c == 0 ? ivx : c == 1 ? ivy : ivz);
            out |= !(s_ > -0.24f && s_ < 1.24f);
        }
    }
    return DIAG && !__any(out);
}

template <bool DIAG, int R, bool MASKED, int PXC>
__global__ __launch_bounds__(256) void rdf_fwd_lane_kernel(
    const float* __restrict__ xyz, int nF, int N, MdgCell cell, float rc2, const uint8_t* __restrict__ mask,
    const float* __restrict__ mu, float coeff, int nbins, const uint32_t* __restrict__ tab, int iters, int pxld,
    float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int rows = nbins + 4 * R;
    const int PXLD = PXC ? PXC : pxld;                          // row stride of the SoA coordinates (even, >= N + 2)
    float* hist = sm + (size_t)wid * ((size_t)rows * 64 + 3 * PXLD);
    float* px = hist + (size_t)rows * 64;                       // [3][PXLD]   (8-byte aligned: rows * 64 and PXLD are even)
    float* smu = sm + (size_t)nw * ((size_t)rows * 64 + 3 * PXLD);   // [nbins + 2R] centres incl. extrapolated pads
    RdfLane<R> K;
    K.init(mu, coeff, nbins);
    for (int k = threadIdx.x; k < nbins + 2 * R; k += blockDim.x) {
        const int kk = k - R;
        smu[k] = (kk >= 0 && kk < nbins) ? mu[kk] : fmaf((float)kk, K.dmu, K.mu0);
    }
    for (int e = lane; e < rows * 64; e += 64) hist[e] = 0.f;
    for (int e = lane; e < 3 * PXLD; e += 64) px[e] = (e >= N && e < PXLD) ? __builtin_nanf("") : 0.f;
    __syncthreads();
    const float ivx = cell.inv[0], ivy = cell.inv[4], ivz = cell.inv[8];
    const int gw = blockIdx.x * nw + wid, nwaves = gridDim.x * nw;
    float* col = hist + lane;
    for (int fr = gw; fr < nF; fr += nwaves) {
        const bool near = rdf_load_frame<DIAG>(px, PXLD, xyz + (size_t)fr * N * 3, N, lane, ivx, ivy, ivz);
        f32x2 dd, mm;                 // distances and nearest centres of the two pairs of the current entry
        int kA, kB;
        bool okA, okB;
        // (measured: a single row of fixed-point counters per wave updated with ds_add_u32 -- 2 KB of LDS
        //  instead of 32 KB, many waves per SIMD -- is 3.5x SLOWER: the lanes of a wave hit the same few
        //  bins around the g(r) peak and the atomics serialise; the lane-private columns never conflict.)
        auto sweep = [&](auto near_c) {
            constexpr bool NEAR = decltype(near_c)::value;
            f32x2 d2;
            int mi = 0, mj = 0;
            auto locate_a = [&](uint32_t t) {
                int i4, j4;
                d2 = rdf_entry_d2<DIAG, NEAR>(px, PXLD, cell, ivx, ivy, ivz, t, i4, j4);
                if constexpr (MASKED) { mi = i4 >> 2; mj = j4 >> 2; }
            };
            auto locate_b = [&]() {
                okA = (d2.x < rc2) & (d2.x != 0.f);                              // (bitwise: no short-circuit branches)
                okB = (d2.y < rc2) & (d2.y != 0.f);
                if constexpr (MASKED) {                                          // unconditional loads: no branch
                    okA = okA & (mask[(size_t)mi * N + min(mj, N - 1)] != 0);
                    okB = okB & (mask[(size_t)mi * N + min(mj + 1, N - 1)] != 0);
                }
                // v_sqrt_f32 (1 ulp): far below the Gaussian's own rounding
                dd = f32x2{__builtin_amdgcn_sqrtf(d2.x), __builtin_amdgcn_sqrtf(d2.y)};
                const f32x2 tk = (dd - K.mu0) * K.inv_dmu;
                kA = (int)rintf(tk.x); kB = (int)rintf(tk.y);
                okA = okA & (kA >= -R) & (kA <= nbins - 1 + R);
                okB = okB & (kB >= -R) & (kB <= nbins - 1 + R);
                kA = okA ? kA : 0; kB = okB ? kB : 0;
                mm = f32x2{smu[kA + R], smu[kB + R]};
            };
            // software pipeline, unrolled by four so that no register rotates.  Each step deposits the entry
            // located by the previous step and locates the one of the next iteration in the shadow of the row
            // reads: first pair's rows | coordinates, squared distances | first deposit | second pair's rows
            // (after the first pair's stores: the two windows may overlap) | distances, bins | second deposit;
            // the table register is reloaded for four iterations later.
            const uint32_t* tp = tab + lane;
            uint32_t t0 = tp[64], t1 = tp[128], t2 = tp[192], t3 = tp[256];
            locate_a(tp[0]);
            locate_b();
            auto step = [&](uint32_t& t, const uint32_t* next) {
                const f32x2 d_ = dd, m_ = mm; const int kb = kB; const bool oa = okA, ob = okB;
                typename RdfLane<R>::Rows w;
                K.rows_load(col, kA, w);
                locate_a(t);
                t = *next;
                K.rows_add_store(w, d_.x, m_.x, oa);
                K.rows_load(col, kb, w);
                locate_b();
                K.rows_add_store(w, d_.y, m_.y, ob);
            };
            for (int it = 0; it < iters; it += 4) {
                tp += 256;
                step(t0, tp + 64);
                step(t1, tp + 128);
                step(t2, tp + 192);
                step(t3, tp + 256);
            }
        };
        if (near) sweep(std::true_type{});
        else sweep(std::false_type{});
    }
    // column sums in a fixed (lane-rotated) order; one partial histogram per wave
    for (int k = lane; k < nbins; k += 64) {
        const float* row = hist + (size_t)(k + 2 * R) * 64;
        float s = 0.f;
        for (int t = 0; t < 64; ++t) s += row[(t + lane) & 63];
        partial[(size_t)gw * nbins + k] = s;
    }
}

// Half-width columns: EIGHT depositing waves per CU (two per SIMD, so one wave's LDS round trips and
// transcendentals hide behind the other's arithmetic).  A wave's histogram is [rows][32]: lanes l and l + 32
// share column l & 31 and split every deposit -- lane l adds the R + 1 rows from the nearest bin upwards, lane
// l + 32 the R + 1 rows below it -- so the two never touch the same row.  Each lane still locates its own table
// entry (two pairs); v_permlane32_swap hands both entries of the lane pair to both lanes (4 instructions for 2 x
// {distance, bin}), and the four pairs are deposited in the same order on both.  One instruction stream for
// both roles: a deposit is R + 1 ascending rows from a per-lane base row, the Gaussian values by the two-term
// recurrence E_(t+1) = E_t rho_t, rho_(t+1) = rho_t c2 (packed: two rows per step, multiplier {rho_t rho_(t+1)}
// advancing by c2^4), anchored at the base row -- for the lower half that is 2^-26 growing towards the centre.
// +46 % VALU instructions per pair against the full-column kernel, at twice the occupancy.
template <bool DIAG, int R, bool MASKED, int PXC>
__global__ __launch_bounds__(512) void rdf_fwd_half_kernel(
    const float* __restrict__ xyz, int nF, int N, MdgCell cell, float rc2, const uint8_t* __restrict__ mask,
    const float* __restrict__ mu, float coeff, int nbins, const uint32_t* __restrict__ tab, int iters, int pxld,
    float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int HR = R + 1;                                   // rows per half deposit
    static_assert(HR % 2 == 0, "rows are handled in adjacent pairs");
    constexpr int PADL = 2 * R + 1;                             // rows below bin 0
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int col = lane & 31;
    const int boff = (lane >> 5) ? -HR : 0;                     // this lane's rows start at bin k + boff
    const int rows = (nbins + PADL + 2 * R + 2) & ~1;
    const int PXLD = PXC ? PXC : pxld;
    float* hist = sm + (size_t)wid * ((size_t)rows * 32 + 3 * PXLD);
    float* px = hist + (size_t)rows * 32;                       // [3][PXLD]
    float* smu = sm + (size_t)nw * ((size_t)rows * 32 + 3 * PXLD);   // [rows] centre of every row
    const float sc = sqrtf(-coeff * LOG2E);
    const float mu0 = mu[0];
    const float dmu = (mu[nbins - 1] - mu0) / (float)(nbins - 1), inv_dmu = 1.f / dmu;
    const float Ds = dmu * sc, Ds2 = Ds * Ds, twoDs = 2.f * Ds;
    const float c2 = __builtin_amdgcn_exp2f(-2.f * Ds2), c4 = (c2 * c2) * (c2 * c2);
    for (int r = threadIdx.x; r < rows; r += blockDim.x) {
        const int kk = r - PADL;
        smu[r] = (kk >= 0 && kk < nbins) ? mu[kk] : fmaf((float)kk, dmu, mu0);
    }
    for (int e = lane; e < rows * 32; e += 64) hist[e] = 0.f;
    for (int e = lane; e < 3 * PXLD; e += 64) px[e] = (e >= N && e < PXLD) ? __builtin_nanf("") : 0.f;
    __syncthreads();
    const float ivx = cell.inv[0], ivy = cell.inv[4], ivz = cell.inv[8];
    const int gw = blockIdx.x * nw + wid, nwaves = gridDim.x * nw;
    float* hcol = hist + col + (size_t)(boff + PADL) * 32;      // row of bin boff
    const float* smub = smu + boff + PADL;
    // R + 1 rows upwards from bin k + boff.  A rejected pair comes in as (1e4, bin 0): y0 clamps to 20, E0
    // underflows to an exact 0 and every product with the (finite) multipliers stays 0.
    // In two parts: the values of all four deposits of a step are computed first, so that the four dependent
    // read-add-write round trips that follow carry one packed add per row pair and nothing else -- that stretch
    // is where the SIMD's other wave runs its arithmetic.
    struct Half { float* h; f32x2 P[HR / 2]; };
    auto half_values = [&](float d, int k, Half& w) {
        w.h = hcol + k * 32;
        const float y0 = fminf((d - smub[k]) * sc, 20.f);
        const float E0 = __builtin_amdgcn_exp2f(-y0 * y0), r0 = __builtin_amdgcn_exp2f(fmaf(y0, twoDs, -Ds2));
        const float r1 = r0 * c2, r2 = r1 * c2;
        f32x2 P = {E0, E0 * r0}, M = f32x2{r0, r1} * f32x2{r1, r2};     // {rho_0 rho_1, rho_1 rho_2}
#pragma unroll
        for (int m_ = 0; m_ < HR / 2; ++m_) {
            w.P[m_] = P;
            if (m_ + 1 < HR / 2) { P *= M; M *= c4; }
        }
    };
    auto half_add = [&](const Half& w) {
        float* h = w.h;
        f32x2 v[HR / 2];
#pragma unroll
        for (int m_ = 0; m_ < HR / 2; ++m_) v[m_] = f32x2{h[(2 * m_) * 32], h[(2 * m_ + 1) * 32]};
#pragma unroll
        for (int m_ = 0; m_ < HR / 2; ++m_) v[m_] += w.P[m_];
#pragma unroll
        for (int m_ = 0; m_ < HR / 2; ++m_) { h[(2 * m_) * 32] = v[m_].x; h[(2 * m_ + 1) * 32] = v[m_].y; }
    };
    for (int fr = gw; fr < nF; fr += nwaves) {
        const bool near = rdf_load_frame<DIAG>(px, PXLD, xyz + (size_t)fr * N * 3, N, lane, ivx, ivy, ivz);
        auto sweep = [&](auto near_c) {
            constexpr bool NEAR = decltype(near_c)::value;
            const uint32_t* tp = tab + lane;
            uint32_t t0 = tp[0], t1 = tp[64], t2 = tp[128], t3 = tp[192];
            // software pipeline over steps: the four read-add-writes of the previous entry pair are spread
            // between the pieces of this one's arithmetic (two register sets, no rotation)
            Half A[4], B[4];
#pragma unroll
            for (int q_ = 0; q_ < 4; ++q_) {
                A[q_].h = hcol;
#pragma unroll
                for (int m_ = 0; m_ < HR / 2; ++m_) A[q_].P[m_] = f32x2{0.f, 0.f};
            }
            auto step = [&](uint32_t& t, const uint32_t* next, const Half (&prev)[4], Half (&cur)[4]) {
                half_add(prev[0]);
                int i4, j4;
                const f32x2 d2 = rdf_entry_d2<DIAG, NEAR>(px, PXLD, cell, ivx, ivy, ivz, t, i4, j4);
                t = *next;
                bool okA = (d2.x < rc2) & (d2.x != 0.f), okB = (d2.y < rc2) & (d2.y != 0.f);
                if constexpr (MASKED) {
                    const int mi = i4 >> 2, mj = j4 >> 2;
                    okA = okA & (mask[(size_t)mi * N + min(mj, N - 1)] != 0);
                    okB = okB & (mask[(size_t)mi * N + min(mj + 1, N - 1)] != 0);
                }
                half_add(prev[1]);
                const f32x2 dd = {__builtin_amdgcn_sqrtf(d2.x), __builtin_amdgcn_sqrtf(d2.y)};
                const f32x2 tk = (dd - mu0) * inv_dmu;
                int kA = (int)rintf(tk.x), kB = (int)rintf(tk.y);
                okA = okA & (kA >= -R) & (kA <= nbins - 1 + R);
                okB = okB & (kB >= -R) & (kB <= nbins - 1 + R);
                const uint32_t uda = __float_as_uint(okA ? dd.x : 1e4f), udb = __float_as_uint(okB ? dd.y : 1e4f);
                const uint32_t uka = (uint32_t)(okA ? kA : 0), ukb = (uint32_t)(okB ? kB : 0);
                // [0]: the value of the lower lane of the pair on both lanes, [1]: the upper lane's
                const auto sda = __builtin_amdgcn_permlane32_swap(uda, uda, false, false);
                const auto sdb = __builtin_amdgcn_permlane32_swap(udb, udb, false, false);
                const auto ska = __builtin_amdgcn_permlane32_swap(uka, uka, false, false);
                const auto skb = __builtin_amdgcn_permlane32_swap(ukb, ukb, false, false);
                half_values(__uint_as_float(sda[0]), (int)ska[0], cur[0]);
                half_values(__uint_as_float(sdb[0]), (int)skb[0], cur[1]);
                half_add(prev[2]);
                half_values(__uint_as_float(sda[1]), (int)ska[1], cur[2]);
                half_add(prev[3]);
                half_values(__uint_as_float(sdb[1]), (int)skb[1], cur[3]);
            };
            for (int it = 0; it < iters; it += 4) {
                tp += 256;
                step(t0, tp, A, B);
                step(t1, tp + 64, B, A);
                step(t2, tp + 128, A, B);
                step(t3, tp + 192, B, A);
            }
#pragma unroll
            for (int q_ = 0; q_ < 4; ++q_) half_add(A[q_]);
        };
        if (near) sweep(std::true_type{});
        else sweep(std::false_type{});
    }
    // column sums in a fixed (lane-rotated) order; one partial histogram per wave
    for (int k = lane; k < nbins; k += 64) {
        const float* row = hist + (size_t)(k + PADL) * 32;
        float s_ = 0.f;
        for (int t = 0; t < 32; ++t) s_ += row[(t + lane) & 31];
        partial[(size_t)gw * nbins + k] = s_;
    }
}

// (measured: the same sweep with the two halves of the work on different waves -- four producer waves turning
//  table entries into distances through a double-buffered LDS ring, four consumer waves in the SIMDs' second
//  slots depositing them, one barrier per four entries -- gives the same bits and is 6 % SLOWER, 18.3 vs 17.3 ms:
//  the wave that deposits is bound by the LDS pipe itself, not by the exposed round trips.  Removed.)

// ---------------------------------------------------------------------------------------------------------
// Fine-grid backward for equally spaced centres (width ~ spacing): dL/dd(d) = sum_k g_k 2 coeff (d - mu_k) e_k(d)
// is ONE smooth function of the distance.  A tiny kernel tabulates it -- value and hf * slope on RDF_SUB nodes
// per centre spacing, x_n = xlo + n hf -- and the per-pair work shrinks from a 13-bin Gaussian sum to the
// evaluation of the cell's cubic Hermite interpolant from LDS (interpolation error ~ (hf/sigma)^4 / 384 * 3
// < 2e-6 of a unit contribution for hf = sigma / 8).
// (The transposed idea for the forward pass -- every pair deposits its four Hermite weights into per-wave
//  node accumulators with fixed-point ds_add_u32, Gaussians evaluated once per wave at the end -- halves the
//  arithmetic but is bound by the LDS atomic unit, ~15 cycles per wave instruction and CU plus 2-3-way address
//  conflicts: 13.5 ms against 11.6 ms for the lane-private columns of rdf_fwd_lane_kernel.  Measured, removed.)
constexpr int RDF_SUB = 8;

struct FineGrid { float xlo, hf, inv_hf; int nn; };

__host__ __device__ inline int fine_nodes(int nbins, int R) { return (nbins - 1 + 2 * (R + 1)) * RDF_SUB + 1; }

// (the centres live in device memory: the grid is derived from them inside the kernels)
__device__ __forceinline__ FineGrid fine_grid(const float* __restrict__ mu, int nbins, int R) {
    FineGrid G;
    const float mu0 = mu[0], dmu = (mu[nbins - 1] - mu0) / (float)(nbins - 1);
    G.xlo = mu0 - (float)(R + 1) * dmu;
    G.hf = dmu / (float)RDF_SUB;
    G.inv_hf = (float)RDF_SUB / dmu;
    G.nn = fine_nodes(nbins, R);
    return G;
}

// node table of dL/dd: tab[n] = (sum_k sg_k x e , hf * d/dd of that), x = s (x_n - mu_k), e = exp2(-x^2)
__device__ __forceinline__ float2 rdf_fine_node(const FineGrid& G, const float* __restrict__ mu, float coeff, int nbins,
                                                const float* __restrict__ g_raw, int R, int n) {
    const float sc = sqrtf(-coeff * LOG2E);
    const float mu0 = mu[0], dmu = (mu[nbins - 1] - mu0) / (float)(nbins - 1);
    const float x = fmaf((float)n, G.hf, G.xlo);
    const int kc = (int)rintf((x - mu0) / dmu);
    float val = 0.f, der = 0.f;
    for (int k = max(0, kc - R - 1); k <= min(nbins - 1, kc + R + 1); ++k) {
        const float xs = (x - mu[k]) * sc;
        const float e = __builtin_amdgcn_exp2f(-xs * xs);
        const float sg = g_raw[k] * 2.f * coeff / sc;
        val = fmaf(sg * xs, e, val);
        der = fmaf(sg * sc * (1.f - 2.f * 0.69314718056f * xs * xs), e, der);
    }
    return make_float2(val, der * G.hf);
}
// cell n = nodes n and n + 1: the cubic Hermite interpolant of {value, hf slope} at both ends, stored as its
// monomial coefficients in the cell coordinate f in [0, 1) (c0 + f (c1 + f (c2 + f c3)): three fma per pair)
__global__ void rdf_bwd_table_kernel(const float* __restrict__ mu, float coeff, int nbins,
                                     const float* __restrict__ g_raw, int R, float4* __restrict__ tab) {
    const FineGrid G = fine_grid(mu, nbins, R);
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= G.nn - 1) return;
    const float2 a_ = rdf_fine_node(G, mu, coeff, nbins, g_raw, R, n), b_ = rdf_fine_node(G, mu, coeff, nbins, g_raw, R, n + 1);
    tab[n] = make_float4(a_.x, a_.y, 3.f * (b_.x - a_.x) - 2.f * a_.y - b_.y, 2.f * (a_.x - b_.x) + a_.y + b_.y);
}

// The u-space variant (rdf_u_grid in common.hpp): node n at u_n = ulo + n hu, r = sqrt(u_n), holds
// W = (dL/dd)(r) / r and hu dW/du = hu (S' / r - S / r^2) / (2 r)  with S = dL/dd, S' = dS/dd.
__global__ void rdf_bwd_table_u_kernel(const float* __restrict__ mu, float coeff, int nbins,
                                       const float* __restrict__ g_raw, int R, float4* __restrict__ tab) {
    float ulo, hu;
    int ncell;
    rdf_u_grid(mu, nbins, R, ulo, hu, ncell);
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= ncell) return;
    const float sc = sqrtf(-coeff * LOG2E);
    const float mu0 = mu[0], dmu = (mu[nbins - 1] - mu0) / (float)(nbins - 1);
    auto node = [&](int m) {
        const float r = sqrtf(fmaf((float)m, hu, ulo));
        const int kc = (int)rintf((r - mu0) / dmu);
        float val = 0.f, der = 0.f;                                  // S and dS/dd, as rdf_fine_node
        for (int k = max(0, kc - R - 1); k <= min(nbins - 1, kc + R + 1); ++k) {
            const float xs = (r - mu[k]) * sc;
            const float e = __builtin_amdgcn_exp2f(-xs * xs);
            const float sg = g_raw[k] * 2.f * coeff / sc;
            val = fmaf(sg * xs, e, val);
            der = fmaf(sg * sc * (1.f - 2.f * 0.69314718056f * xs * xs), e, der);
        }
        const float ir = 1.0f / r;
        return make_float2(val * ir, hu * (der * ir - val * ir * ir) * (0.5f * ir));
    };
    const float2 a_ = node(n), b_ = node(n + 1);
    tab[n] = make_float4(a_.x, a_.y, 3.f * (b_.x - a_.x) - 2.f * a_.y - b_.y, 2.f * (a_.x - b_.x) + a_.y + b_.y);
}

// Fine-grid backward, one wave per frame.  Pair order: lane <-> atom i (64 at a time), step s <-> partner
// j = (i + s) mod N, s = 1 .. N/2 (for an even N the last step only for i < N/2): every unordered pair once, and
// in one step the partners of the 64 lanes are all different, so the partners' gradient read-add-writes never
// collide, while atom i's own gradient accumulates in registers (half the LDS traffic of a pair order that
// scatters both ends).  Positions and partner gradients live in index-doubled arrays (entry k and k + N are the
// same atom; the two gradient halves are folded at the end): the partner address is affine in s, no modulo.
// Two steps per iteration in packed fp32; the stores of step s precede the reads of step s + 1 (lane l's second
// partner is lane l + 1's first).  The table holds one float4 per cell: the cell's cubic in monomial form.
template <bool DIAG>
__global__ __launch_bounds__(256) void rdf_bwd_fine_kernel(
    const float* __restrict__ xyz, int nF, int N, MdgCell cell, float rc2, const uint8_t* __restrict__ mask,
    const float* __restrict__ mu, int nbins, int R, const float4* __restrict__ tab_g, int LDW,
    float* __restrict__ g_xyz) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const FineGrid G = fine_grid(mu, nbins, R);
    float4* tab = reinterpret_cast<float4*>(sm);                     // [nn - 1] cells
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float* px = sm + 4 * (size_t)(G.nn - 1) + (size_t)wid * 6 * LDW; // positions [3][LDW] then gradient [3][LDW]
    float* gx = px + 3 * LDW;
    for (int n = threadIdx.x; n < G.nn - 1; n += blockDim.x) tab[n] = tab_g[n];
    for (int e = lane; e < 6 * LDW; e += 64) px[e] = 0.f;
    __syncthreads();
    const int fr = blockIdx.x * (blockDim.x >> 6) + wid;
    if (fr >= nF) return;
    const float* pos = xyz + (size_t)fr * N * 3;
    const float ivx = cell.inv[0], ivy = cell.inv[4], ivz = cell.inv[8];
    bool out = false;
    for (int e = lane; e < 3 * N; e += 64) {
        const int c = e % 3, a = e / 3;
        const float v = pos[e];
        px[c * LDW + a] = v;
        px[c * LDW + a + N] = v;
        if (DIAG) {
            const float s_ = v * (c == 0 ? ivx : c == 1 ? ivy : ivz);
            out |= !(s_ > -0.24f && s_ < 1.24f);
        }
    }
    const bool near = DIAG && !__any(out);                           // see rdf_fwd_lane_kernel
    const float tmax = (float)(G.nn - 1);
    const int smax = N / 2;                                          // steps (the last one is the half step of an even N)
    auto sweep = [&](auto near_c) {
        constexpr bool NEAR = decltype(near_c)::value;
        for (int i0 = 0; i0 < N; i0 += 64) {
            const int i = i0 + lane;                                 // (lanes past N run on zeros and add +-0)
            const bool live = i < N;
            const int lim = !live ? 0 : ((N & 1) ? (N - 1) / 2 : (i >= N / 2 ? smax - 1 : smax));
            const float xi = px[i], yi = px[LDW + i], zi = px[2 * LDW + i];
            f32x2 ax = {0.f, 0.f}, ay = ax, az = ax;                 // - gradient of atom i, two partial sums
            const float* pj = px + i;
            float* gj = gx + i;
            for (int s_ = 1; s_ <= smax; s_ += 2) {
                pj += 2; gj += 2;                                    // -> partner i + s_ + 1 (the first one is at [-1])
                f32x2 dx = f32x2{pj[-1], pj[0]} - xi, dy = f32x2{pj[LDW - 1], pj[LDW]} - yi,
                      dz = f32x2{pj[2 * LDW - 1], pj[2 * LDW]} - zi;   // D = x_j - x_i
                if constexpr (NEAR) {
                    dx = min_image_diag2_near(dx, ivx, cell.h[0]);
                    dy = min_image_diag2_near(dy, ivy, cell.h[4]);
                    dz = min_image_diag2_near(dz, ivz, cell.h[8]);
                } else if constexpr (DIAG) {
                    dx = min_image_diag2(dx, ivx, cell.h[0]);
                    dy = min_image_diag2(dy, ivy, cell.h[4]);
                    dz = min_image_diag2(dz, ivz, cell.h[8]);
                } else {
                    float ax_ = dx.x, ay_ = dy.x, az_ = dz.x, bx_ = dx.y, by_ = dy.y, bz_ = dz.y;
                    min_image<false>(cell, ax_, ay_, az_);
                    min_image<false>(cell, bx_, by_, bz_);
                    dx = f32x2{ax_, bx_}; dy = f32x2{ay_, by_}; dz = f32x2{az_, bz_};
                }
                const f32x2 d2 = norm2_ref2(dx, dy, dz);
                bool okA = (s_ <= lim) & (d2.x < rc2) & (d2.x != 0.f), okB = (s_ + 1 <= lim) & (d2.y < rc2) & (d2.y != 0.f);
                if (mask) {                                                  // (uniform)
                    const int ii = live ? i : 0;
                    int ja = ii + s_, jb = ii + s_ + 1;
                    ja = ja >= N ? ja - N : ja; jb = jb >= N ? jb - N : jb;
                    jb = jb >= N ? jb - N : jb;
                    okA = okA & (mask[(size_t)min(ii, ja) * N + max(ii, ja)] != 0);
                    okB = okB & (mask[(size_t)min(ii, jb) * N + max(ii, jb)] != 0);
                }
                const f32x2 id = {__builtin_amdgcn_rsqf(okA ? d2.x : 1.f), __builtin_amdgcn_rsqf(okB ? d2.y : 1.f)};
                f32x2 t = (d2 * id - G.xlo) * G.inv_hf;
                okA = okA & (t.x >= 0.f) & (t.x < tmax);
                okB = okB & (t.y >= 0.f) & (t.y < tmax);
                t = f32x2{okA ? t.x : 0.f, okB ? t.y : 0.f};
                const int gA = (int)t.x, gB = (int)t.y;
                const f32x2 f = t - f32x2{(float)gA, (float)gB};
                const float4 ca = tab[gA], cb = tab[gB];
                const float sdA = fmaf(f.x, fmaf(f.x, fmaf(f.x, ca.w, ca.z), ca.y), ca.x);
                const float sdB = fmaf(f.y, fmaf(f.y, fmaf(f.y, cb.w, cb.z), cb.y), cb.x);
                // d(dist)/dx_j = +D/d, d(dist)/dx_i = -D/d; a rejected pair adds +-0
                const f32x2 cw = f32x2{okA ? sdA : 0.f, okB ? sdB : 0.f} * id;
                const f32x2 cx = cw * dx, cy = cw * dy, cz = cw * dz;
                ax += cx; ay += cy; az += cz;
                {
                    const float g0 = gj[-1], g1 = gj[LDW - 1], g2 = gj[2 * LDW - 1];
                    gj[-1] = g0 + cx.x; gj[LDW - 1] = g1 + cy.x; gj[2 * LDW - 1] = g2 + cz.x;
                }
                {
                    const float g0 = gj[0], g1 = gj[LDW], g2 = gj[2 * LDW];
                    gj[0] = g0 + cx.y; gj[LDW] = g1 + cy.y; gj[2 * LDW] = g2 + cz.y;
                }
            }
            if (live) {
                gx[i] -= ax.x + ax.y; gx[LDW + i] -= ay.x + ay.y; gx[2 * LDW + i] -= az.x + az.y;
            }
        }
    };
    if (near) sweep(std::true_type{});
    else sweep(std::false_type{});
    float* out_g = g_xyz + (size_t)fr * N * 3;
    for (int e = lane; e < 3 * N; e += 64) {
        const int c = e % 3, a = e / 3;
        out_g[e] = gx[c * LDW + a] + gx[c * LDW + a + N];
    }
}

// ---------------------------------------------------------------------------------------------
// Forward for many frames and equally spaced centres through a FINE INTEGER histogram.
// The soft histogram  raw[k] = sum_pairs exp2(-(s (d - mu_k))^2)  is a smooth function of every pair distance, so
// the pairs are first COUNTED on a grid M times finer than the centres (one integer LDS atomic per pair: a
// `ds_add_u32` instead of the 11 float read-add-writes of the lane-private columns), and the fine counts are smeared
// onto the centres once at the end:  raw[k] = sum_m H[m] exp2(-(s (x_m - mu_k))^2),  x_m = centre of fine bin m.
// Error: moving a pair to its bin centre changes its Gaussian by (delta^2 / 2) G'' on average (the first-order
// term averages out over the bin): relative (s h)^2 [(2 ln2 u)^2 - 2 ln2] / 24 at u = s |d - mu|, i.e. 9e-7 at the
// peak and 1.1e-5 three widths out (where the Gaussian is 2^-9 of its peak) for s h = 0.004 -- the tests allow 2e-5
// per bin.  Integer counts commute, so the result is independent of the order in which waves and workgroups
// deposit: bitwise reproducible without lane-private columns -- the sixteen waves of a workgroup share ONE histogram
// (~24 000 fine bins = 95 KB at 100 centres) and the workgroups merge into a global integer histogram.
// Pairs come from a table (i | j << 16), row-major in (i, j): the lanes of one step share atom i (LDS broadcast) and
// read consecutive j (conflict-free), and masked (species-selected) histograms simply have a shorter table.
constexpr int RDF_FINE_MAX = 36864;       // fine bins that fit the workgroup's LDS histogram (144 KB)
constexpr int RDF_FINE_WAVES = 16;
constexpr int RDF_FINE_ATOMS = 1024;

__global__ __launch_bounds__(1024) void rdf_fine_table_kernel(int N, const uint8_t* __restrict__ mask,
                                                              uint32_t* __restrict__ tab, int32_t* __restrict__ count) {
    __shared__ int32_t cnt[RDF_FINE_ATOMS + 1];
    const int i = threadIdx.x;
    int c = 0;
    if (i < N) {
        if (mask) { for (int j = i + 1; j < N; ++j) c += mask[(size_t)i * N + j] != 0; }
        else c = N - 1 - i;
    }
    cnt[i] = c;
    __syncthreads();
    if (threadIdx.x == 0) {                          // N <= 1024: a serial scan is a few microseconds
        int run = 0;
        for (int k = 0; k < N; ++k) { const int v = cnt[k]; cnt[k] = run; run += v; }
        cnt[N] = run;
        *count = run;
    }
    __syncthreads();
    if (i < N) {
        int o = cnt[i];
        for (int j = i + 1; j < N; ++j)                   // entries hold BYTE offsets (4 i | 4 j << 16): no shifts in the loop
            if (!mask || mask[(size_t)i * N + j]) tab[o++] = (uint32_t)(4 * i) | ((uint32_t)(4 * j) << 16);
    }
    // padding up to the next multiple of 128 entries: the pair (0, 0) has distance 0, which lies below the fine grid
    const int total = cnt[N], padded = (total + 127) & ~127;
    for (int o = total + threadIdx.x; o < padded; o += blockDim.x) tab[o] = 0u;
}

// One frame of one wave.  SAFE: the fine grid starts above 0 and ends inside the cutoff (checked once per kernel), so
// `0 <= t < nfine` alone accepts exactly the pairs of topology.py:67 that can contribute -- a zero distance (padding
// entries, coincident atoms) falls below the grid, a distance beyond the cutoff above it.
template <bool DIAG, bool SAFE>
__device__ __forceinline__ void rdf_fine_frame(const MdgCell& cell, const uint32_t* __restrict__ tab, int P, float rc2,
                                               float inv_h, float tlo, float fmax, const float* px, const float* py,
                                               const float* pz, uint32_t* hist, int lane) {
    const char* bx = reinterpret_cast<const char*>(px);
    const char* by = reinterpret_cast<const char*>(py);
    const char* bz = reinterpret_cast<const char*>(pz);
#define LDF(base, off) (*reinterpret_cast<const float*>((base) + (off)))
    if constexpr (DIAG) {
        // two table entries per lane and step in packed fp32 (the same image arithmetic as min_image<true>)
        const float iv0 = cell.inv[0], iv1 = cell.inv[4], iv2 = cell.inv[8];
        const float h0 = cell.h[0], h1 = cell.h[4], h2 = cell.h[8];
        for (int p0 = 0; p0 < P; p0 += 128) {
            const uint32_t ea = tab[p0 + lane], eb = tab[p0 + 64 + lane];
            const int ia = (int)(ea & 0xFFFFu), ja = (int)(ea >> 16), ib = (int)(eb & 0xFFFFu), jb = (int)(eb >> 16);
            f32x2 dx = f32x2{LDF(bx, ja), LDF(bx, jb)} - f32x2{LDF(bx, ia), LDF(bx, ib)};
            f32x2 dy = f32x2{LDF(by, ja), LDF(by, jb)} - f32x2{LDF(by, ia), LDF(by, ib)};
            f32x2 dz = f32x2{LDF(bz, ja), LDF(bz, jb)} - f32x2{LDF(bz, ia), LDF(bz, ib)};
            dx = min_image_diag2(dx, iv0, h0); dy = min_image_diag2(dy, iv1, h1); dz = min_image_diag2(dz, iv2, h2);
            const f32x2 d2 = norm2_ref2(dx, dy, dz);
            const float ta = fmaf(__builtin_amdgcn_sqrtf(d2.x), inv_h, tlo), tb = fmaf(__builtin_amdgcn_sqrtf(d2.y), inv_h, tlo);
            bool oka = ta >= 0.f && ta < fmax, okb = tb >= 0.f && tb < fmax;
            if (!SAFE) { oka = oka && d2.x < rc2 && d2.x != 0.f; okb = okb && d2.y < rc2 && d2.y != 0.f; }
            if (oka) atomicAdd(&hist[(int)ta], 1u);
            if (okb) atomicAdd(&hist[(int)tb], 1u);
        }
    } else {
        for (int p0 = 0; p0 < P; p0 += 64) {
            const uint32_t ent = tab[p0 + lane];
            const int i = (int)(ent & 0xFFFFu), j = (int)(ent >> 16);
            float dx = LDF(bx, j) - LDF(bx, i), dy = LDF(by, j) - LDF(by, i), dz = LDF(bz, j) - LDF(bz, i);
            min_image<DIAG>(cell, dx, dy, dz);
            const float d2 = norm2_ref(dx, dy, dz);
            const float t = fmaf(__builtin_amdgcn_sqrtf(d2), inv_h, tlo);
            bool ok = t >= 0.f && t < fmax;
            if (!SAFE) ok = ok && d2 < rc2 && d2 != 0.f;
            if (ok) atomicAdd(&hist[(int)t], 1u);
        }
    }
#undef LDF
}

template <bool DIAG>
__global__ __launch_bounds__(64 * RDF_FINE_WAVES) void rdf_fwd_fine_kernel(
    const float* __restrict__ xyz, int nF, int N, MdgCell cell, float rc2, const uint32_t* __restrict__ tab,
    const int32_t* __restrict__ count, const float* __restrict__ mu, float reach, float inv_h, int nfine, int ld,
    uint32_t* __restrict__ ghist) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    const float lo = mu[0] - reach;                             // lower edge of the fine grid
    uint32_t* hist = reinterpret_cast<uint32_t*>(smf);          // [nfine] shared by the workgroup's waves
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float* px = smf + nfine + wid * 3 * ld;                     // wave-private SoA frame
    float* py = px + ld;
    float* pz = py + ld;
    for (int m = threadIdx.x; m < nfine; m += blockDim.x) hist[m] = 0u;
    __syncthreads();
    const int P = *count;                                       // (the table is padded to a multiple of 128 entries)
    const float fmax = (float)nfine, tlo = -lo * inv_h;
    const float hi = lo + fmax / inv_h;
    const bool safe = lo > 0.f && hi * hi <= rc2;
    for (int fr = blockIdx.x * RDF_FINE_WAVES + wid; fr < nF; fr += gridDim.x * RDF_FINE_WAVES) {
        const float* pos = xyz + (size_t)fr * N * 3;
        for (int e = lane; e < 3 * N; e += 64) {                // AoS -> SoA (coalesced read)
            const int a = e / 3, c = e - 3 * a;
            px[c * ld + a] = pos[e];
        }
        // (px is private to the wave: program order + the LDS counter suffice)
        if (safe) rdf_fine_frame<DIAG, true>(cell, tab, P, rc2, inv_h, tlo, fmax, px, py, pz, hist, lane);
        else rdf_fine_frame<DIAG, false>(cell, tab, P, rc2, inv_h, tlo, fmax, px, py, pz, hist, lane);
    }
    __syncthreads();
    for (int m = threadIdx.x; m < nfine; m += blockDim.x) {
        const uint32_t v = hist[m];
        if (v) atomicAdd(&ghist[m], v);
    }
}

// The same fine histogram fed from a neighbour list (large systems, few frames: config #4's 4 096-atom liquid):
// thread per (atom, slot) of the per-atom list, every pair counted once (slot's neighbour index above the atom's).
// Frames are stacked as groups of the list (mdg_nbr_build_cell_groups), so one launch covers all of them.
__global__ __launch_bounds__(1024) void rdf_fwd_ell_kernel(
    const float* __restrict__ pos, long long n_slots, MdgCell cell, const int32_t* __restrict__ col,
    const int32_t* __restrict__ shift, const int32_t* __restrict__ cnt, int max_nbr, const float* __restrict__ mu,
    float reach, float inv_h, int nfine, uint32_t* __restrict__ ghist) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    uint32_t* hist = reinterpret_cast<uint32_t*>(smf);
    for (int m = threadIdx.x; m < nfine; m += blockDim.x) hist[m] = 0u;
    __syncthreads();
    const float lo = mu[0] - reach, tlo = -lo * inv_h, fmax = (float)nfine;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n_slots; t += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(t / max_nbr), k = (int)(t - (long long)i * max_nbr);
        if (k >= cnt[i]) continue;
        const int j = col[t];
        if (j <= i) continue;
        float dx = pos[3 * i] - pos[3 * j], dy = pos[3 * i + 1] - pos[3 * j + 1], dz = pos[3 * i + 2] - pos[3 * j + 2];
        apply_shift(cell, shift[t], dx, dy, dz);
        const float tt = fmaf(__builtin_amdgcn_sqrtf(norm2_ref(dx, dy, dz)), inv_h, tlo);
        if (tt >= 0.f && tt < fmax) atomicAdd(&hist[(int)tt], 1u);
    }
    __syncthreads();
    for (int m = threadIdx.x; m < nfine; m += blockDim.x) {
        const uint32_t v = hist[m];
        if (v) atomicAdd(&ghist[m], v);
    }
}

// raw[k] = sum_m H[m] exp2(-(s (x_m - mu_k))^2): one wave per centre over the fine bins within reach
__global__ void rdf_fine_finish_kernel(const uint32_t* __restrict__ ghist, int nfine, float h,
                                       const float* __restrict__ mu, float sc, float reach, int nbins,
                                       float* __restrict__ raw) {
    const int k = blockIdx.x, lane = threadIdx.x;
    const float m_k = mu[k], lo = mu[0] - reach;
    int m0 = (int)floorf((m_k - reach - lo) / h), m1 = (int)ceilf((m_k + reach - lo) / h);
    m0 = m0 < 0 ? 0 : m0;
    m1 = m1 > nfine ? nfine : m1;
    double s = 0.0;
    for (int m = m0 + lane; m < m1; m += 64) {
        const uint32_t c = ghist[m];
        if (c) {
            const float x = (lo + ((float)m + 0.5f) * h - m_k) * sc;
            s += (double)c * (double)__builtin_amdgcn_exp2f(-x * x);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) raw[k] = (float)s;
}

// raw[k] = sum_b partial[b][k] in fixed order; one wave per bin
__global__ void rdf_finish_kernel(const float* __restrict__ partial, int nblocks, int nbins,
                                  float* __restrict__ raw) {
    const int k = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int b = lane; b < nblocks; b += 64) s += partial[(size_t)b * nbins + k];
    s = wave_sum(s);
    if (lane == 0) raw[k] = s;
}

// Backward: one WAVE per frame, every unordered pair visited once.  The pairs are walked in
// round-robin-tournament order (circle method): round r holds floor(N'/2) DISJOINT pairs, so
// the lanes of one instruction never touch the same atom and the +/- contributions go into a
// wave-private LDS gradient with plain read-modify-writes -- no atomics, and the rounds are
// sequential within the wave => fixed summation order, bitwise reproducible.
// R > 0: equally spaced centres, dL/dd from the 2R+1 bins around the nearest centre with the forward
// kernel's recurrence (tables padded so that no bin needs a bounds test); R == 0: direct sum, `uniform`
// only narrows the bin range.
template <bool DIAG, int R>
__global__ __launch_bounds__(256) void rdf_bwd_kernel(
    const float* __restrict__ xyz, int nF, int N, MdgCell cell, float rc2, const uint8_t* __restrict__ mask,
    const float* __restrict__ mu, float coeff, int nbins, const float* __restrict__ g_raw,
    float* __restrict__ g_xyz, int uniform) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* smu = sm;                         // [nbins + 2R]  centre of bin k at k + R
    float* sg = sm + nbins + 2 * R;          // [nbins + 4R]  g_k * 2 coeff / s at k + 2R, zero padding
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float* px = sg + nbins + 4 * R + (size_t)wid * 6 * N;   // positions [3][N] then gradient [3][N] of this wave's frame
    float* gx = px + 3 * N;
    const float sc = sqrtf(-coeff * LOG2E);
    {
        const float m0 = mu[0], dm = nbins > 1 ? (mu[nbins - 1] - m0) / (float)(nbins - 1) : 0.f;
        for (int k = threadIdx.x; k < nbins + 2 * R; k += blockDim.x) {
            const int kk = k - R;
            smu[k] = (kk >= 0 && kk < nbins) ? mu[kk] : fmaf((float)kk, dm, m0);
        }
        for (int k = threadIdx.x; k < nbins + 4 * R; k += blockDim.x) {
            const int kk = k - 2 * R;
            sg[k] = (kk >= 0 && kk < nbins) ? g_raw[kk] * 2.f * coeff / sc : 0.f;
        }
    }
    __syncthreads();
    const int fr = blockIdx.x * (blockDim.x >> 6) + wid;
    if (fr >= nF) return;
    const float* pos = xyz + (size_t)fr * N * 3;
    for (int e = lane; e < 3 * N; e += 64) { px[(e % 3) * N + e / 3] = pos[e]; gx[e] = 0.f; }
    const float mu0 = smu[R];
    const float dmu = nbins > 1 ? (smu[R + nbins - 1] - mu0) / (float)(nbins - 1) : 0.f;
    const float reach = 11.3f / sc;          // exp2 argument below -126 beyond this
    const float inv_dmu = dmu > 0.f ? 1.f / dmu : 0.f;
    const float Ds = dmu * sc, Ds2 = Ds * Ds;
    const float c2 = __builtin_amdgcn_exp2f(-2.f * Ds2);
    const int Np = N + (N & 1);              // even number of tournament slots (one dummy when N is odd)
    const int M = Np - 1, half = Np / 2;
    for (int r = 0; r < M; ++r) {
        for (int p0 = 0; p0 < half; p0 += 64) {
            const int p = p0 + lane;
            int a = -1, b = -1;
            if (p < half) {
                if (p == 0) { a = M; b = r; }
                else { a = r + p; if (a >= M) a -= M; b = r - p; if (b < 0) b += M; }
                if (a >= N || b >= N) a = -1;                     // pair with the dummy slot
            }
            if (a >= 0) {
                const int i = min(a, b), j = max(a, b);
                float dx = px[j] - px[i], dy = px[N + j] - px[N + i], dz = px[2 * N + j] - px[2 * N + i];
                min_image<DIAG>(cell, dx, dy, dz);               // D = x_j - x_i as the forward pass
                const float d2 = norm2_ref(dx, dy, dz);
                bool ok = (d2 < rc2) && (d2 != 0.f);
                if (ok && mask) ok = mask[(size_t)i * N + j] != 0;
                if (ok) {
                    const float id = __builtin_amdgcn_rsqf(d2);
                    const float d = d2 * id;
                    // dL/dd = sum_k g_k 2 coeff (d - mu_k) e_k = sum_k sg_k x_k exp2(-x_k^2), x_k = s (d - mu_k)
                    float sd = 0.f;
                    if constexpr (R > 0) {
                        const int kc = (int)rintf((d - mu0) * inv_dmu);
                        if (kc >= -R && kc <= nbins - 1 + R) {
                            const float x0 = (d - smu[kc + R]) * sc;
                            const float a = 2.f * Ds * x0;
                            float eu = __builtin_amdgcn_exp2f(-x0 * x0), ed = eu, xu = x0, xd = x0;
                            float ru = __builtin_amdgcn_exp2f(a - Ds2), rd = __builtin_amdgcn_exp2f(-a - Ds2);
                            const float* gk = sg + kc + R;        // gk[R] is bin kc
                            sd = gk[R] * x0 * eu;
#pragma unroll
                            for (int s = 1; s <= R; ++s) {
                                eu *= ru; ru *= c2; xu -= Ds; sd = fmaf(gk[R + s] * xu, eu, sd);
                                ed *= rd; rd *= c2; xd += Ds; sd = fmaf(gk[R - s] * xd, ed, sd);
                            }
                        }
                    } else {
                        int klo = 0, khi = nbins - 1;
                        if (uniform && dmu > 0.f) {
                            klo = max(0, (int)floorf((d - reach - mu0) / dmu));
                            khi = min(nbins - 1, (int)ceilf((d + reach - mu0) / dmu));
                        }
                        for (int k = klo; k <= khi; ++k) {
                            const float x = (d - smu[k]) * sc;
                            sd = fmaf(sg[k] * x, __builtin_amdgcn_exp2f(-x * x), sd);
                        }
                    }
                    const float c = sd * id;                      // d(dist)/dx_j = +D/d, d(dist)/dx_i = -D/d
                    gx[j] += c * dx; gx[N + j] += c * dy; gx[2 * N + j] += c * dz;
                    gx[i] -= c * dx; gx[N + i] -= c * dy; gx[2 * N + i] -= c * dz;
                }
            }
        }
    }
    float* out = g_xyz + (size_t)fr * N * 3;
    for (int e = lane; e < 3 * N; e += 64) out[e] = gx[(e % 3) * N + e / 3];
}

// Backward, few-frames / large-N variant: LPA lanes per (frame, atom) gather (each pair visited from
// both ends; used when there are too few frames to fill the chip with one wave per frame).
template <bool DIAG, int LPA>
__global__ void rdf_bwd_atom_kernel(const float* __restrict__ xyz, int nF, int N, MdgCell cell, float rc2,
                               const uint8_t* __restrict__ mask, const float* __restrict__ mu, float coeff,
                               int nbins, const float* __restrict__ g_raw, float* __restrict__ g_xyz, int uniform) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* smu = sm;             // [nbins]
    float* sg = sm + nbins;      // [nbins]
    const float sc = sqrtf(-coeff * LOG2E);
    for (int k = threadIdx.x; k < nbins; k += blockDim.x) { smu[k] = mu[k]; sg[k] = g_raw[k] * 2.f * coeff / sc; }
    __syncthreads();
    const int apb = blockDim.x / LPA;
    const long long gi = (long long)blockIdx.x * apb + threadIdx.x / LPA;
    const int sub = threadIdx.x % LPA;
    if (gi >= (long long)nF * N) return;
    const int fr = (int)(gi / N), i = (int)(gi % N);
    const float* pos = xyz + (size_t)fr * N * 3;
    const float xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
    // bins whose Gaussian is non-negligible (exp2 argument > -126) around a distance
    const float dmu = nbins > 1 ? (smu[nbins - 1] - smu[0]) / (float)(nbins - 1) : 0.f;
    const float reach = 11.3f / sc;                                       // sqrt(126) in scaled units
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int j = sub; j < N; j += LPA) {
        if (j == i) continue;
        float dx = pos[3 * j] - xi, dy = pos[3 * j + 1] - yi, dz = pos[3 * j + 2] - zi;
        // the half list holds (min(i,j), max(i,j)) with D = x_hi - x_lo: evaluate in that
        // orientation so the accepted set is the forward pass's
        const bool flip = j < i;
        if (flip) { dx = -dx; dy = -dy; dz = -dz; }
        min_image<DIAG>(cell, dx, dy, dz);
        const float d2 = norm2_ref(dx, dy, dz);
        if (!((d2 < rc2) && (d2 != 0.f))) continue;
        if (mask && !mask[(size_t)(flip ? j : i) * N + (flip ? i : j)]) continue;
        const float id = __builtin_amdgcn_rsqf(d2);
        const float ds = d2 * id;
        int klo = 0, khi = nbins - 1;
        if (uniform && dmu > 0.f) {
            klo = max(0, (int)floorf((ds - reach - smu[0]) / dmu));
            khi = min(nbins - 1, (int)ceilf((ds + reach - smu[0]) / dmu));
        }
        // dL/dd = sum_k g_k 2 coeff (d - mu_k) e_k = sum_k sg_k x_k exp2(-x_k^2),
        //   x_k = s (d - mu_k), sg_k = g_k * 2 coeff / s
        float s = 0.f;
        for (int k = klo; k <= khi; ++k) {
            const float x = (ds - smu[k]) * sc;
            s = fmaf(sg[k] * x, __builtin_amdgcn_exp2f(-x * x), s);
        }
        // d(dist)/dx_i = -(D)/d for D = x_j - x_i (unflipped); with flip, D was negated
        const float c = (flip ? s : -s) * id;
        gx = fmaf(c, dx, gx); gy = fmaf(c, dy, gy); gz = fmaf(c, dz, gz);
    }
    gx = group_sum<LPA>(gx); gy = group_sum<LPA>(gy); gz = group_sum<LPA>(gz);
    if (sub == 0) {
        float* o = g_xyz + ((size_t)fr * N + i) * 3;
        o[0] = gx; o[1] = gy; o[2] = gz;
    }
}

}  // namespace

static int rdf_grid(int n_frames, int n_atoms) {
    const long long npair = (long long)n_atoms * (n_atoms - 1) / 2;
    const long long items = (long long)n_frames * ((npair + RDF_CHUNK - 1) / RDF_CHUNK);
    return (int)(items < RDF_MAX_BLOCKS ? items : RDF_MAX_BLOCKS);
}

extern "C" int64_t mdg_rdf_partial_size(int n_frames, int n_atoms, int nbins) {
    (void)n_frames; (void)n_atoms;
    return (int64_t)RDF_MAX_BLOCKS * nbins;      // one partial histogram per persistent block / wave
}

// lane-per-pair kernel: reach R (bins) for the scaled spacing Ds, waves per workgroup that fit the LDS
static int rdf_lane_reach(float spacing_s) {
    // the nearest uncovered bin is R + 1/2 spacings from the distance: (R + 1/2) Ds >= 5.3 => below 2^-28
    if (spacing_s >= 0.82f) return 6;
    if (spacing_s >= 0.465f) return 11;
    return 0;
}
// reach of the forward lane kernel (the 2R+1 rows a pair touches are its LDS traffic, so it is cut closer than
// the backward table's): (R + 1/2) Ds >= 4.65
static int rdf_fwd_reach(float spacing_s) {
    if (spacing_s >= 0.8455f) return 5;
    if (spacing_s >= 0.405f) return 11;
    return 0;
}
static size_t rdf_lane_lds(int nw, int R, int pxld, int nbins) {
    return sizeof(float) * ((size_t)(nbins + 2 * R) + (size_t)nw * ((size_t)(nbins + 4 * R) * 64 + 3 * (size_t)pxld));
}

static int rdf_fwd_impl(const float* xyz, int n_frames, int n_atoms, const MdgCell* cell, float cutoff,
                        const uint8_t* mask, const float* mu, float coeff, int nbins, float* raw,
                        float* partial, float spacing_s, void* stream) {
    MDG_CHECK_ARG(xyz && cell && mu && raw && partial, "rdf_fwd: null buffer");
    MDG_CHECK_ARG(n_frames > 0 && n_atoms > 1 && nbins > 0, "rdf_fwd: bad sizes");
    MDG_CHECK_ARG(nbins <= RDF_BLOCK, "rdf_fwd: nbins > %d not supported", RDF_BLOCK);
    MDG_CHECK_ARG(coeff < 0.f, "rdf_fwd: coeff must be negative (-0.5 / width^2)");
    const int nblocks = rdf_grid(n_frames, n_atoms);
    hipStream_t st = (hipStream_t)stream;
    // equally spaced centres with Ds = s * spacing <= 1: 8-bin blocks + recurrence (spacing_s is the
    // caller's statement that mu is a linspace; <= 0 selects the direct kernel)
    // ---- many frames, equally spaced centres: fine integer histogram (see rdf_fwd_fine_kernel) when it fits the LDS
    // (the first-order term of the binning error averages out statistically, ~ (s h) 0.4 / sqrt(pairs near a centre):
    //  small systems -- cheap anyway -- keep the exact lane-private kernels below)
    if (n_frames >= 1024 && nbins >= 2 && spacing_s > 0.f && n_atoms >= 64 && n_atoms <= RDF_FINE_ATOMS) {
        const float sc = sqrtf(-coeff * LOG2E);                   // exp(coeff x^2) = exp2(-(sc x)^2)
        const float spacing = spacing_s / sc;
        const float reach = 5.3f / sc;                            // beyond: below 2^-28 of the peak
        const float h = 0.004f / sc;                              // sc h = 0.004 (error bound: see the kernel)
        const double span = (double)(nbins - 1) * spacing + 2.0 * reach;
        const long long nfine = (long long)ceil(span / h) + 1;
        const int ld = (n_atoms + 1) & ~1;
        const size_t lds = sizeof(float) * ((size_t)nfine + RDF_FINE_WAVES * 3 * (size_t)ld);
        if (nfine <= RDF_FINE_MAX && lds <= 156 * 1024) {
            const long long npair = (long long)n_atoms * (n_atoms - 1) / 2;
            uint32_t* scratch = nullptr;                          // [nfine] global histogram | count | pair table
            const size_t words = (size_t)nfine + 4 + (size_t)npair + 128;
            if (hipMallocAsync((void**)&scratch, sizeof(uint32_t) * words, st) == hipSuccess && scratch) {
                uint32_t* ghist = scratch;
                int32_t* count = reinterpret_cast<int32_t*>(scratch + nfine);
                uint32_t* tab = scratch + nfine + 4;
                MDG_HIP(hipMemsetAsync(ghist, 0, sizeof(uint32_t) * (size_t)(nfine + 4), st));
                hipLaunchKernelGGL(rdf_fine_table_kernel, dim3(1), dim3(1024), 0, st, n_atoms, mask, tab, count);
                int grid = (n_frames + RDF_FINE_WAVES - 1) / RDF_FINE_WAVES;
                if (grid > 256) grid = 256;                       // one resident workgroup (16 waves) per CU
                // (the fine grid starts at mu[0] - reach; the kernels read mu[0] themselves: no host copy)
                if (cell->diag)
                    hipLaunchKernelGGL(rdf_fwd_fine_kernel<true>, dim3(grid), dim3(64 * RDF_FINE_WAVES), lds, st, xyz, n_frames, n_atoms,
                                       *cell, cutoff * cutoff, tab, count, mu, reach, 1.0f / h, (int)nfine, ld, ghist);
                else
                    hipLaunchKernelGGL(rdf_fwd_fine_kernel<false>, dim3(grid), dim3(64 * RDF_FINE_WAVES), lds, st, xyz, n_frames, n_atoms,
                                       *cell, cutoff * cutoff, tab, count, mu, reach, 1.0f / h, (int)nfine, ld, ghist);
                hipLaunchKernelGGL(rdf_fine_finish_kernel, dim3(nbins), dim3(64), 0, st, ghist, (int)nfine, h, mu, sc, reach,
                                   nbins, raw);
                (void)hipFreeAsync(scratch, st);
                MDG_CHECK_LAUNCH("rdf_fwd_fine_kernel");
                return MDG_OK;
            }
        }
    }
    const int R = n_frames >= 1024 && nbins >= 2 && n_atoms <= RDF_TABLE_MAX_ATOMS ? rdf_fwd_reach(spacing_s) : 0;
    if (R) {
        // coordinate stride: even, with at least two NaN columns after the atoms for the padding entries;
        // compile-time (immediate LDS offsets) for the common shape: orthorhombic, no mask
        const int npad_col = (n_atoms + 1) & ~1;
        const bool px128 = npad_col + 2 <= 128 && cell->diag && !mask;
        const int pxld = px128 ? 128 : npad_col + 2;
        int nw = 4;
        while (nw > 1 && rdf_lane_lds(nw, R, pxld, nbins) > 156 * 1024) nw >>= 1;
        const size_t lds = rdf_lane_lds(nw, R, pxld, nbins);
        int n_entries = 0;
        for (int i = 0; i + 1 < n_atoms; ++i) n_entries += rdf_row_entries(i, n_atoms);
        const int iters = (((n_entries + 63) / 64) + 3) & ~3;               // (the loop is unrolled by four)
        const int n_padded = (iters + 5) * 64;
        const uint32_t pad_entry = (uint32_t)(4 * npad_col) | ((uint32_t)(4 * npad_col) << 16);
        uint32_t* tab = nullptr;
        if (lds <= 156 * 1024 && hipMallocAsync((void**)&tab, sizeof(uint32_t) * (size_t)n_padded, st) == hipSuccess && tab) {
            hipLaunchKernelGGL(rdf_pair_table_kernel, dim3(n_atoms), dim3(256), 0, st, tab, n_atoms, n_entries, n_padded,
                               pad_entry);
            // half-width columns, eight waves per workgroup, when they fit (R = 5, 100 bins: 137 KB)
            const size_t rows_h = (size_t)((nbins + 4 * R + 3) & ~1);
            const size_t lds_half = sizeof(float) * (rows_h + 8 * (rows_h * 32 + 3 * (size_t)pxld));
            // (spacing_s <= 1.5: the recurrence multipliers of a rejected pair, y0 clamped to 20, stay finite)
            if (R == 5 && spacing_s <= 1.5f && lds_half <= 156 * 1024) {
                int gridh = (n_frames + 7) / 8;
                if (gridh > RDF_MAX_BLOCKS / 8) gridh = RDF_MAX_BLOCKS / 8;
#define MDG_RDF_HALF(D, MK, PX)                                                                                        \
    hipLaunchKernelGGL((rdf_fwd_half_kernel<D, 5, MK, PX>), dim3(gridh), dim3(512), lds_half, st, xyz, n_frames,       \
                       n_atoms, *cell, cutoff * cutoff, mask, mu, coeff, nbins, tab, iters, pxld, partial)
                if (px128) MDG_RDF_HALF(true, false, 128);
                else if (cell->diag && mask) MDG_RDF_HALF(true, true, 0);
                else if (cell->diag) MDG_RDF_HALF(true, false, 0);
                else if (mask) MDG_RDF_HALF(false, true, 0);
                else MDG_RDF_HALF(false, false, 0);
#undef MDG_RDF_HALF
                hipLaunchKernelGGL(rdf_finish_kernel, dim3(nbins), dim3(64), 0, st, partial, gridh * 8, nbins, raw);
                (void)hipFreeAsync(tab, st);
                MDG_CHECK_LAUNCH("rdf_fwd_half_kernel");
                return MDG_OK;
            }
            int grid = (n_frames + nw - 1) / nw;
            if (grid > RDF_MAX_BLOCKS / nw) grid = RDF_MAX_BLOCKS / nw;
#define MDG_RDF_LANE(D, RR, MK, PX)                                                                                    \
    hipLaunchKernelGGL((rdf_fwd_lane_kernel<D, RR, MK, PX>), dim3(grid), dim3(64 * nw), lds, st, xyz, n_frames,        \
                       n_atoms, *cell, cutoff * cutoff, mask, mu, coeff, nbins, tab, iters, pxld, partial)
#define MDG_RDF_LANE_R(RR)                                             \
    do {                                                               \
        if (px128) MDG_RDF_LANE(true, RR, false, 128);                 \
        else if (cell->diag && mask) MDG_RDF_LANE(true, RR, true, 0);  \
        else if (cell->diag) MDG_RDF_LANE(true, RR, false, 0);         \
        else if (mask) MDG_RDF_LANE(false, RR, true, 0);               \
        else MDG_RDF_LANE(false, RR, false, 0);                        \
    } while (0)
            if (R == 5) MDG_RDF_LANE_R(5); else MDG_RDF_LANE_R(11);
#undef MDG_RDF_LANE_R
#undef MDG_RDF_LANE
            hipLaunchKernelGGL(rdf_finish_kernel, dim3(nbins), dim3(64), 0, st, partial, grid * nw, nbins, raw);
            (void)hipFreeAsync(tab, st);
            MDG_CHECK_LAUNCH("rdf_fwd_lane_kernel");
            return MDG_OK;
        }
        if (tab) (void)hipFreeAsync(tab, st);
    }
    const int nblk = (nbins + RDF_KB - 1) / RDF_KB;
    const bool block8 = spacing_s > 0.f && spacing_s <= 1.0f && nbins >= 2 * RDF_KB && nblk <= RDF_BLOCK;
    if (block8) {
        const int G = RDF_BLOCK / nblk;
        const size_t lds = sizeof(float) * (RDF_BLOCK + 16 + (size_t)G * nblk * RDF_KB);
        if (cell->diag)
            hipLaunchKernelGGL(rdf_fwd_block8_kernel<true>, dim3(nblocks), dim3(RDF_BLOCK), lds, st, xyz, n_frames,
                               n_atoms, *cell, cutoff * cutoff, mask, mu, coeff, nbins, partial);
        else
            hipLaunchKernelGGL(rdf_fwd_block8_kernel<false>, dim3(nblocks), dim3(RDF_BLOCK), lds, st, xyz, n_frames,
                               n_atoms, *cell, cutoff * cutoff, mask, mu, coeff, nbins, partial);
    } else if (cell->diag)
        hipLaunchKernelGGL(rdf_fwd_kernel<true>, dim3(nblocks), dim3(RDF_BLOCK), 0, st, xyz, n_frames, n_atoms,
                           *cell, cutoff * cutoff, mask, mu, coeff, nbins, partial);
    else
        hipLaunchKernelGGL(rdf_fwd_kernel<false>, dim3(nblocks), dim3(RDF_BLOCK), 0, st, xyz, n_frames, n_atoms,
                           *cell, cutoff * cutoff, mask, mu, coeff, nbins, partial);
    hipLaunchKernelGGL(rdf_finish_kernel, dim3(nbins), dim3(64), 0, st, partial, nblocks, nbins, raw);
    MDG_CHECK_LAUNCH("rdf_fwd_kernel");
    return MDG_OK;
}

extern "C" int mdg_rdf_fwd(const float* xyz, int n_frames, int n_atoms, const MdgCell* cell, float cutoff,
                           const uint8_t* mask, const float* mu, float coeff, int nbins, float* raw,
                           float* partial, void* stream) {
    return rdf_fwd_impl(xyz, n_frames, n_atoms, cell, cutoff, mask, mu, coeff, nbins, raw, partial, 0.f, stream);
}

extern "C" int mdg_rdf_fwd_uniform(const float* xyz, int n_frames, int n_atoms, const MdgCell* cell, float cutoff,
                                   const uint8_t* mask, const float* mu, float spacing, float coeff, int nbins,
                                   float* raw, float* partial, void* stream) {
    const float spacing_s = spacing > 0.f && coeff < 0.f ? spacing * sqrtf(-coeff * 1.4426950408889634f) : 0.f;
    return rdf_fwd_impl(xyz, n_frames, n_atoms, cell, cutoff, mask, mu, coeff, nbins, raw, partial, spacing_s, stream);
}

static int rdf_bwd_impl(const float* xyz, int n_frames, int n_atoms, const MdgCell* cell, float cutoff,
                        const uint8_t* mask, const float* mu, float coeff, int nbins, const float* g_raw,
                        float* g_xyz, float spacing_s, void* stream) {
    MDG_CHECK_ARG(xyz && cell && mu && g_raw && g_xyz, "rdf_bwd: null buffer");
    MDG_CHECK_ARG(n_frames > 0 && n_atoms > 1 && nbins > 0, "rdf_bwd: bad sizes");
    MDG_CHECK_ARG(coeff < 0.f, "rdf_bwd: coeff must be negative (-0.5 / width^2)");
    hipStream_t st = (hipStream_t)stream;
    const int uniform = spacing_s > 0.f;
    // few frames or frames too large for LDS: (frame, atom) gather variant
    if (n_frames < 1024 || n_atoms > 4096) {
        constexpr int LPA = 16, BLK = 256;
        const long long rows = (long long)n_frames * n_atoms;
        const int nb = (int)((rows + BLK / LPA - 1) / (BLK / LPA));
        const size_t l2 = sizeof(float) * 2 * nbins;
        if (cell->diag)
            hipLaunchKernelGGL((rdf_bwd_atom_kernel<true, LPA>), dim3(nb), dim3(BLK), l2, st, xyz, n_frames, n_atoms,
                               *cell, cutoff * cutoff, mask, mu, coeff, nbins, g_raw, g_xyz, uniform);
        else
            hipLaunchKernelGGL((rdf_bwd_atom_kernel<false, LPA>), dim3(nb), dim3(BLK), l2, st, xyz, n_frames, n_atoms,
                               *cell, cutoff * cutoff, mask, mu, coeff, nbins, g_raw, g_xyz, uniform);
        MDG_CHECK_LAUNCH("rdf_bwd_atom_kernel");
        return MDG_OK;
    }
    const int R = nbins >= 2 ? rdf_lane_reach(spacing_s) : 0;
    if (R && spacing_s <= 1.0f && fine_nodes(nbins, R) <= 4096) {
        // table of dL/dd on the fine nodes (stream-ordered scratch: no state, re-entrant), then the
        // wave-per-frame kernel with a table lookup per pair
        const int nn = fine_nodes(nbins, R);
        const size_t tabf = 4 * (size_t)(nn - 1);
        // index-doubled rows: the last lane group's idle lanes reach index 64 ceil(N/64) - 1 + N/2 + 1
        const int ldw = (max(2 * n_atoms, 64 * ((n_atoms + 63) / 64) + n_atoms / 2 + 1) + 2) & ~1;
        int wpb = 4;
        while (wpb > 1 && sizeof(float) * (tabf + (size_t)wpb * 6 * ldw) > 150 * 1024) wpb >>= 1;
        const size_t lds = sizeof(float) * (tabf + (size_t)wpb * 6 * ldw);
        if (lds <= 160 * 1024) {
            float4* tab = nullptr;
            if (hipMallocAsync((void**)&tab, sizeof(float4) * (size_t)(nn - 1), st) != hipSuccess || !tab) {
                mdg_set_error("rdf_bwd: scratch allocation failed");
                return MDG_ELAUNCH;
            }
            hipLaunchKernelGGL(rdf_bwd_table_kernel, dim3((nn + 254) / 256), dim3(256), 0, st, mu, coeff, nbins, g_raw, R, tab);
            const int nblocks = (n_frames + wpb - 1) / wpb;
            if (cell->diag)
                hipLaunchKernelGGL(rdf_bwd_fine_kernel<true>, dim3(nblocks), dim3(64 * wpb), lds, st, xyz, n_frames, n_atoms,
                                   *cell, cutoff * cutoff, mask, mu, nbins, R, tab, ldw, g_xyz);
            else
                hipLaunchKernelGGL(rdf_bwd_fine_kernel<false>, dim3(nblocks), dim3(64 * wpb), lds, st, xyz, n_frames, n_atoms,
                                   *cell, cutoff * cutoff, mask, mu, nbins, R, tab, ldw, g_xyz);
            (void)hipFreeAsync(tab, st);
            MDG_CHECK_LAUNCH("rdf_bwd_fine_kernel");
            return MDG_OK;
        }
    }
    // waves (= frames) per workgroup limited by the 6N floats of LDS each one needs
    const size_t tables = 2 * (size_t)nbins + 6 * (size_t)R;
    int wpb = 4;
    while (wpb > 1 && sizeof(float) * (tables + (size_t)wpb * 6 * n_atoms) > 150 * 1024) wpb >>= 1;
    const size_t lds = sizeof(float) * (tables + (size_t)wpb * 6 * n_atoms);
    MDG_CHECK_ARG(lds <= 160 * 1024, "rdf_bwd: N=%d does not fit the LDS-resident frame kernel", n_atoms);
    const int nblocks = (n_frames + wpb - 1) / wpb;
#define MDG_RDF_BWD(D, RR)                                                                                        \
    hipLaunchKernelGGL((rdf_bwd_kernel<D, RR>), dim3(nblocks), dim3(64 * wpb), lds, st, xyz, n_frames, n_atoms, *cell, \
                       cutoff * cutoff, mask, mu, coeff, nbins, g_raw, g_xyz, uniform)
    if (cell->diag) { if (R == 6) MDG_RDF_BWD(true, 6); else if (R == 11) MDG_RDF_BWD(true, 11); else MDG_RDF_BWD(true, 0); }
    else            { if (R == 6) MDG_RDF_BWD(false, 6); else if (R == 11) MDG_RDF_BWD(false, 11); else MDG_RDF_BWD(false, 0); }
#undef MDG_RDF_BWD
    MDG_CHECK_LAUNCH("rdf_bwd_kernel");
    return MDG_OK;
}

extern "C" int mdg_rdf_bwd(const float* xyz, int n_frames, int n_atoms, const MdgCell* cell, float cutoff,
                           const uint8_t* mask, const float* mu, float coeff, int nbins, const float* g_raw,
                           float* g_xyz, void* stream) {
    return rdf_bwd_impl(xyz, n_frames, n_atoms, cell, cutoff, mask, mu, coeff, nbins, g_raw, g_xyz, 0.f, stream);
}

extern "C" int mdg_rdf_bwd_uniform(const float* xyz, int n_frames, int n_atoms, const MdgCell* cell, float cutoff,
                                   const uint8_t* mask, const float* mu, float spacing, float coeff, int nbins,
                                   const float* g_raw, float* g_xyz, void* stream) {
    const float spacing_s = spacing > 0.f && coeff < 0.f ? spacing * sqrtf(-coeff * 1.4426950408889634f) : 0.f;
    return rdf_bwd_impl(xyz, n_frames, n_atoms, cell, cutoff, mask, mu, coeff, nbins, g_raw, g_xyz, spacing_s, stream);
}

// fine grid of the integer-histogram forward for centres mu0 + k spacing and exp(coeff x^2): number of fine bins, or
// 0 when it does not fit the workgroup's LDS histogram
static long long rdf_fine_bins(float spacing, float coeff, int nbins, float* reach, float* h) {
    if (!(spacing > 0.f) || !(coeff < 0.f) || nbins < 2) return 0;
    const float sc = sqrtf(-coeff * LOG2E);
    *reach = 5.3f / sc;
    *h = 0.004f / sc;
    const double span = (double)(nbins - 1) * spacing + 2.0 * (*reach);
    const long long nfine = (long long)ceil(span / *h) + 1;
    return nfine <= RDF_FINE_MAX ? nfine : 0;
}

// ---- shared with the fused observable of the trajectory kernels (common.hpp)
RdfFinePlan mdg_rdf_fine_plan(float spacing, float coeff, int nbins) {
    RdfFinePlan P{};
    if (!(spacing > 0.f) || !(coeff < 0.f) || nbins < 2) return P;
    P.sc = sqrtf(-coeff * LOG2E);
    P.nfine = rdf_fine_bins(spacing, coeff, nbins, &P.reach, &P.h);
    const float spacing_s = spacing * P.sc;
    const int R = rdf_lane_reach(spacing_s);
    if (R && spacing_s <= 1.0f && fine_nodes(nbins, R) <= 4096) { P.reach_bins = R; P.ncell = fine_nodes(nbins, R) - 1; }
    return P;
}
int mdg_rdf_fine_finish(const uint32_t* ghist, const RdfFinePlan& P, const float* mu, int nbins, float* raw, hipStream_t st) {
    hipLaunchKernelGGL(rdf_fine_finish_kernel, dim3(nbins), dim3(64), 0, st, ghist, (int)P.nfine, P.h, mu, P.sc, P.reach,
                       nbins, raw);
    MDG_CHECK_LAUNCH("rdf_fine_finish_kernel");
    return MDG_OK;
}
int mdg_rdf_bwd_table(const float* mu, float coeff, int nbins, const float* g_raw, const RdfFinePlan& P, float4* tab,
                      hipStream_t st) {
    const int nn = P.ncell + 1;
    hipLaunchKernelGGL(rdf_bwd_table_kernel, dim3((nn + 254) / 256), dim3(256), 0, st, mu, coeff, nbins, g_raw, P.reach_bins, tab);
    MDG_CHECK_LAUNCH("rdf_bwd_table_kernel");
    return MDG_OK;
}

int mdg_rdf_bwd_table_u(const float* mu, float coeff, int nbins, const float* g_raw, const RdfFinePlan& P, float4* tab,
                        hipStream_t st) {
    hipLaunchKernelGGL(rdf_bwd_table_u_kernel, dim3((P.ncell + 255) / 256), dim3(256), 0, st, mu, coeff, nbins, g_raw,
                       P.reach_bins, tab);
    MDG_CHECK_LAUNCH("rdf_bwd_table_u_kernel");
    return MDG_OK;
}

extern "C" int mdg_rdf_ell_supported(float spacing, float coeff, int nbins) {
    float reach, h;
    return rdf_fine_bins(spacing, coeff, nbins, &reach, &h) > 0;
}

extern "C" int mdg_rdf_fwd_ell(const float* pos, int64_t n_atoms_total, const MdgCell* cell, const int32_t* col,
                               const int32_t* shift, const int32_t* cnt, int max_nbr, const float* mu, float spacing,
                               float coeff, int nbins, float* raw, void* stream) {
    MDG_CHECK_ARG(pos && cell && col && shift && cnt && mu && raw && n_atoms_total > 0 && max_nbr > 0, "rdf_fwd_ell: bad arguments");
    float reach, h;
    const long long nfine = rdf_fine_bins(spacing, coeff, nbins, &reach, &h);
    MDG_CHECK_ARG(nfine > 0, "rdf_fwd_ell: the fine grid for these centres does not fit (see mdg_rdf_ell_supported)");
    hipStream_t st = (hipStream_t)stream;
    uint32_t* ghist = nullptr;
    MDG_HIP(hipMallocAsync((void**)&ghist, sizeof(uint32_t) * (size_t)nfine, st));
    MDG_HIP(hipMemsetAsync(ghist, 0, sizeof(uint32_t) * (size_t)nfine, st));
    const long long n_slots = (long long)n_atoms_total * max_nbr;
    long long grid = (n_slots + 1023) / 1024;
    if (grid > 256) grid = 256;
    const float sc = sqrtf(-coeff * LOG2E);
    hipLaunchKernelGGL(rdf_fwd_ell_kernel, dim3((unsigned)grid), dim3(1024), sizeof(uint32_t) * (size_t)nfine, st, pos, n_slots,
                       *cell, col, shift, cnt, max_nbr, mu, reach, 1.0f / h, (int)nfine, ghist);
    hipLaunchKernelGGL(rdf_fine_finish_kernel, dim3(nbins), dim3(64), 0, st, ghist, (int)nfine, h, mu, sc, reach, nbins, raw);
    (void)hipFreeAsync(ghist, st);
    MDG_CHECK_LAUNCH("rdf_fwd_ell_kernel");
    return MDG_OK;
}
