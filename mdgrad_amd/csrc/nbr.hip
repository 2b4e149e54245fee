// K1: neighbour list builders (replace generate_nbr_list, torchmd/topology.py:30-73).
//
// Output layout (see include/mdgrad_hip.h): padded per-atom FULL list, rows sorted by
// neighbour index -- the layout the per-atom gather kernels want (no scatter, no atomics);
// the reference's lexicographic half list is the (j > i) subsequence of every row.
//
//   dense : one wave per row scans all j in index order; ordered compaction with
//           ballot/popcount keeps rows sorted.  Any cell (triclinic ok).
//   cell  : bin atoms (wrapped fractional coordinates) -> counting sort -> one wave per
//           atom walks the 27-bin stencil, then rank-sorts its row in LDS.  The pair test is
//           the SAME arithmetic as the dense path (reference minimum image + un-contracted
//           d^2), so both produce identical lists.
#include <stdarg.h>
#include <string.h>
#include "common.hpp"

static thread_local char g_err[512] = "";
void mdg_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* mdg_last_error(void) { return g_err; }
extern "C" int mdg_version(void) { return 100; }

namespace {

__device__ __forceinline__ unsigned long long lanemask_lt() {
    return (1ull << (threadIdx.x & 63)) - 1ull;
}

// pair test shared by both builders: returns image code, or -1 when (i,j) is not a neighbour
template <bool DIAG>
__device__ __forceinline__ int pair_test(const MdgCell& c, const float* __restrict__ pos, int i, int j,
                                         float xi, float yi, float zi, float rc2,
                                         const uint8_t* __restrict__ mask, int N) {
    float dx = pos[3 * j] - xi, dy = pos[3 * j + 1] - yi, dz = pos[3 * j + 2] - zi;
    const int code = min_image<DIAG>(c, dx, dy, dz);
    const float d2 = norm2_ref(dx, dy, dz);
    bool ok = (d2 < rc2) && (d2 != 0.f);
    if (ok && mask) ok = mask[(size_t)i * N + j] != 0;
    return ok ? code : -1;
}

// `group` = atoms per independent replica (atoms i and j interact only inside one group of
// consecutive indices; the mask, when given, is [group, group]); group == N for a single system.
template <bool DIAG>
__global__ void nbr_dense_kernel(const float* __restrict__ pos, int N, int group, MdgCell cell, float rc2,
                                 const uint8_t* __restrict__ mask, int32_t* __restrict__ col,
                                 int32_t* __restrict__ shift, int32_t* __restrict__ cnt, int max_nbr,
                                 int32_t* __restrict__ overflow, const int32_t* __restrict__ gate = nullptr,
                                 float* __restrict__ pos_build = nullptr, int32_t* __restrict__ row_half = nullptr) {
    if (gate && gate[0] == 0) return;                            // (Verlet reuse: the stored list still holds)
    const int lane = threadIdx.x & 63;
    const int i = xcd_chunk(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);   // XCD-aware atom order
    if (i >= N) return;
    const float xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
    if (pos_build && lane < 3) pos_build[3 * i + lane] = pos[3 * i + lane];      // where this list was built
    const int g0 = (i / group) * group, g1 = min(N, g0 + group);
    int base = 0, half = 0;
    for (int j0 = g0; j0 < g1; j0 += 64) {
        const int j = j0 + lane;
        int code = -1;
        if (j < g1 && j != i) {
            code = pair_test<DIAG>(cell, pos, i, j, xi, yi, zi, rc2, nullptr, N);
            if (code >= 0 && mask && !mask[(size_t)(i - g0) * group + (j - g0)]) code = -1;
        }
        const unsigned long long b = __ballot(code >= 0);
        bool upper = false;
        if (code >= 0) {
            const int k = base + __popcll(b & lanemask_lt());
            if (k < max_nbr) { col[(size_t)i * max_nbr + k] = j; shift[(size_t)i * max_nbr + k] = code; upper = j > i; }
        }
        half += __popcll(__ballot(upper));
        base += __popcll(b);
    }
    if (lane == 0) {
        cnt[i] = base < max_nbr ? base : max_nbr;
        if (row_half) row_half[i] = half;                        // stored entries with j > i (what half_count_kernel finds)
        if (base > max_nbr) atomicMax(overflow, base);
    }
}

// ---------------------------------------------------------------------------- cell list
struct Bins { int nb[3]; int ncell; };

__host__ __device__ inline Bins make_bins(const MdgCell& c, float cutoff) {
    Bins b;
    for (int d = 0; d < 3; ++d) {
        int n = (int)floorf(c.h[4 * d] / cutoff);
        b.nb[d] = n < 1 ? 1 : n;
    }
    b.ncell = b.nb[0] * b.nb[1] * b.nb[2];
    return b;
}

__device__ __forceinline__ int bin_coord(float x, float inv, int nb) {
    float fr = x * inv;
    fr -= floorf(fr);
    int b = (int)(fr * (float)nb);
    return b >= nb ? nb - 1 : (b < 0 ? 0 : b);
}

// (replica-stacked systems: every group of `group` consecutive atoms has its own set of bins, group g owning
//  bins [g ncell, (g+1) ncell) -- pairs never cross groups)
__global__ void bin_count_kernel(const float* __restrict__ pos, int N, int group, MdgCell cell, Bins bins,
                                 int32_t* __restrict__ atom_bin, int32_t* __restrict__ bin_cnt,
                                 const int32_t* __restrict__ gate = nullptr, float* __restrict__ pos_build = nullptr) {
    if (gate && gate[0] == 0) return;                            // (Verlet reuse: the stored list still holds)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (pos_build) { pos_build[3 * i] = pos[3 * i]; pos_build[3 * i + 1] = pos[3 * i + 1]; pos_build[3 * i + 2] = pos[3 * i + 2]; }
    const int bx = bin_coord(pos[3 * i], cell.inv[0], bins.nb[0]);
    const int by = bin_coord(pos[3 * i + 1], cell.inv[4], bins.nb[1]);
    const int bz = bin_coord(pos[3 * i + 2], cell.inv[8], bins.nb[2]);
    const int b = (i / group) * bins.ncell + (bx * bins.nb[1] + by) * bins.nb[2] + bz;
    atom_bin[i] = b;
    atomicAdd(&bin_cnt[b], 1);
}

// single-block exclusive scan: out[k] = sum_{l<k} in[l], out[n] = total; also copies to cursor.  `out` may alias `in`.
// Every wave owns one contiguous segment: pass 1 sums it (coalesced, independent loads), one barrier and a 16-entry prefix
// give the segment offsets, pass 2 re-reads the segment in 64-item chunks (L2 hits) and writes the running wave scan --
// two barriers whatever n (the chunk-by-chunk version it replaces took 3 block barriers per 1 024 items: 37 us for the
// 32 768 row counts of eight stacked 4 096-bead replicas).
__global__ void scan_kernel(const int32_t* in, int n, int32_t* out, int32_t* __restrict__ cursor,
                            const int32_t* __restrict__ gate = nullptr) {
    if (gate && gate[0] == 0) return;                            // (Verlet reuse: the stored list still holds)
    __shared__ int32_t wsum[16];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int seg = ((n + nw - 1) / nw + 63) / 64 * 64;            // items per wave, a multiple of the chunk
    const int b = min(n, wid * seg), e = min(n, b + seg);
    int s = 0;
    for (int k = b + lane; k < e; k += 64) s += in[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) wsum[wid] = s;
    __syncthreads();                                             // (every read of pass 1 precedes every write of pass 2)
    int run = 0;
    for (int w = 0; w < wid; ++w) run += wsum[w];
    for (int k0 = b; k0 < e; k0 += 64) {
        const int k = k0 + lane;
        const int v = k < e ? in[k] : 0;
        int x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
        if (k < e) { out[k] = run + x - v; if (cursor) cursor[k] = run + x - v; }
        run += __shfl(x, 63, 64);
    }
    if (wid == nw - 1 && lane == 0) out[n] = run;                  // (segments past n are empty: run = the total)
}

__global__ void bin_fill_kernel(const int32_t* __restrict__ atom_bin, int N, int32_t* __restrict__ cursor,
                                int32_t* __restrict__ sorted_atoms, const int32_t* __restrict__ gate = nullptr) {
    if (gate && gate[0] == 0) return;                            // (Verlet reuse: the stored list still holds)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int slot = atomicAdd(&cursor[atom_bin[i]], 1);
    sorted_atoms[slot] = i;
}

// Counting sort of one group's atoms into its bins by ONE workgroup (counts, scan and cursors in LDS): what
// zero + bin_count_kernel + scan_kernel + bin_fill_kernel do in four launches, for groups whose bins fit LDS_BINS.
// A group owns atoms [g group, (g+1) group) of sorted_atoms, so bin_start needs no scan across groups.  The order of the
// atoms inside a bin depends on the LDS atomics; the rows built from it are rank-sorted, so the list does not.
constexpr int LDS_BINS = 8192;
constexpr int SORT_BLOCK = 1024;
constexpr int SORT_MAX_GROUP = 1 << 16;
__global__ __launch_bounds__(SORT_BLOCK) void bin_sort_group_kernel(
    const float* __restrict__ pos, int N, int group, MdgCell cell, Bins bins, int32_t* __restrict__ atom_bin,
    int32_t* __restrict__ bin_start, int32_t* __restrict__ sorted_atoms, const int32_t* __restrict__ gate,
    float* __restrict__ pos_build) {
    if (gate && gate[0] == 0) return;                            // (Verlet reuse: the stored list still holds)
    extern __shared__ __attribute__((aligned(16))) int32_t sm[];
    __shared__ int32_t wsum[SORT_BLOCK / 64];
    const int nc = bins.ncell, g = blockIdx.x, g0 = g * group;
    int32_t* c = sm;                                             // counts, then cursors
    int32_t* st = sm + nc;                                       // exclusive starts (local to the group)
    for (int b = threadIdx.x; b < nc; b += SORT_BLOCK) c[b] = 0;
    __syncthreads();
    for (int a = threadIdx.x; a < group; a += SORT_BLOCK) {
        const int i = g0 + a;
        const float x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
        if (pos_build) { pos_build[3 * i] = x; pos_build[3 * i + 1] = y; pos_build[3 * i + 2] = z; }
        const int b = (bin_coord(x, cell.inv[0], bins.nb[0]) * bins.nb[1] + bin_coord(y, cell.inv[4], bins.nb[1])) * bins.nb[2] +
                      bin_coord(z, cell.inv[8], bins.nb[2]);
        atom_bin[i] = g * nc + b;
        atomicAdd(&c[b], 1);
    }
    __syncthreads();
    // exclusive scan of the counts: every thread owns `per` consecutive bins
    const int per = (nc + SORT_BLOCK - 1) / SORT_BLOCK, b0 = min(nc, (int)threadIdx.x * per), b1 = min(nc, b0 + per);
    int mine = 0;
    for (int b = b0; b < b1; ++b) mine += c[b];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int x = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    int run = x - mine;
    for (int w = 0; w < wid; ++w) run += wsum[w];
    for (int b = b0; b < b1; ++b) { const int n = c[b]; st[b] = run; run += n; }
    __syncthreads();
    for (int b = threadIdx.x; b < nc; b += SORT_BLOCK) { c[b] = st[b]; bin_start[g * nc + b] = g0 + st[b]; }
    if (g == (int)gridDim.x - 1 && threadIdx.x == 0) bin_start[(size_t)gridDim.x * nc] = N;
    __syncthreads();
    for (int a = threadIdx.x; a < group; a += SORT_BLOCK) {
        const int i = g0 + a;
        const int slot = atomicAdd(&c[atom_bin[i] - g * nc], 1);
        sorted_atoms[g0 + slot] = i;
    }
}

constexpr int ROW_CAP = 512;   // LDS row buffer per wave (entries)

__global__ void nbr_cell_kernel(const float* __restrict__ pos, int N, int group, MdgCell cell, Bins bins, float rc2,
                                const uint8_t* __restrict__ mask, const int32_t* __restrict__ atom_bin,
                                const int32_t* __restrict__ bin_start, const int32_t* __restrict__ sorted_atoms,
                                int32_t* __restrict__ col, int32_t* __restrict__ shift,
                                int32_t* __restrict__ cnt, int max_nbr, int32_t* __restrict__ overflow,
                                const int32_t* __restrict__ gate = nullptr, int32_t* __restrict__ row_half = nullptr) {
    if (gate && gate[0] == 0) return;                            // (Verlet reuse: the stored list still holds)
    extern __shared__ __attribute__((aligned(16))) int32_t sm[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int32_t* bj = sm + wid * 2 * ROW_CAP;
    int32_t* bc = bj + ROW_CAP;
    const int i = blockIdx.x * (blockDim.x >> 6) + wid;
    if (i >= N) return;
    const float xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
    const int g0 = (i / group) * group, gbin = (i / group) * bins.ncell;
    const int b = atom_bin[i] - gbin;
    const int bz = b % bins.nb[2], by = (b / bins.nb[2]) % bins.nb[1], bx = b / (bins.nb[2] * bins.nb[1]);
    int base = 0;
    // the three z-neighbours of a stencil column are consecutive bins, i.e. ONE contiguous range of the sorted atom
    // array (two when the column wraps around the cell): 9-11 ranges of ~3 bins instead of 27 single bins -- the
    // rows are rank-sorted below, so the visiting order does not matter.  The ranges are laid end to end and the wave walks
    // the concatenation 64 candidates at a time (round 6): a range of ~24 atoms per pass left 60 % of the lanes idle (ten
    // passes of the pair test per atom where four cover the ~220 candidates; list build at 8 x 4 096 beads 137 -> 81 us,
    // profiles/r06_nbr_kbench.txt).
    // Lane s < 18 holds range s: start, length; their running sum comes from a wave scan.
    const int nbz = bins.nb[2];
    int r_a0 = 0, r_len = 0;
    if (lane < 18) {
        const int s = lane, colm = s >> 1, part = s & 1;
        const int cx = (bx + colm / 3 - 1 + bins.nb[0]) % bins.nb[0];
        const int cy = (by + colm % 3 - 1 + bins.nb[1]) % bins.nb[1];
        const int cb = gbin + (cx * bins.nb[1] + cy) * nbz;
        int zlo = bz - 1, zhi = bz + 1;                       // inclusive, before wrapping
        bool on = true;
        if (part == 0) { zlo = max(zlo, 0); zhi = min(zhi, nbz - 1); }
        else if (bz == 0) { zlo = zhi = nbz - 1; }            // wrapped remainder
        else if (bz == nbz - 1) { zlo = zhi = 0; }
        else on = false;
        if (on) { r_a0 = bin_start[cb + zlo]; r_len = bin_start[cb + zhi + 1] - r_a0; }
    }
    int r_end = r_len;                                         // inclusive scan over lanes 0..17 (lanes >= 18 hold zeros)
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        const int v = __shfl_up(r_end, o, 64);
        if (lane >= o) r_end += v;
    }
    const int r_shift = r_a0 - (r_end - r_len);                // flat position q of range s sits at sorted index q + shift[s]
    int ends[18], shifts[18];                                  // (wave-uniform: scalar registers)
#pragma unroll
    for (int s = 0; s < 18; ++s) {
        ends[s] = __builtin_amdgcn_readlane(r_end, s);
        shifts[s] = __builtin_amdgcn_readlane(r_shift, s);
    }
    const int total = ends[17];
    for (int q0 = 0; q0 < total; q0 += 64) {
        const int q = q0 + lane;
        int sh = shifts[0];
#pragma unroll
        for (int s = 0; s < 17; ++s) sh = q >= ends[s] ? shifts[s + 1] : sh;      // (an empty range: end[s] = end[s - 1], passed over)
        int code = -1, j = -1;
        if (q < total) {
            j = sorted_atoms[q + sh];
            if (j != i) {
                code = pair_test<true>(cell, pos, i, j, xi, yi, zi, rc2, nullptr, N);
                if (code >= 0 && mask && !mask[(size_t)(i - g0) * group + (j - g0)]) code = -1;
            }
        }
        const unsigned long long bal = __ballot(code >= 0);
        if (code >= 0) {
            const int k = base + __popcll(bal & lanemask_lt());
            if (k < ROW_CAP) { bj[k] = j; bc[k] = code; }
        }
        base += __popcll(bal);
    }
    const int n = base < ROW_CAP ? base : ROW_CAP;
    // rank sort by neighbour index (entries are distinct)
    int half = 0;
    for (int k = lane; k < n; k += 64) {
        const int jk = bj[k];
        int rank = 0;
        for (int l = 0; l < n; ++l) rank += bj[l] < jk;
        if (rank < max_nbr) { col[(size_t)i * max_nbr + rank] = jk; shift[(size_t)i * max_nbr + rank] = bc[k]; half += jk > i; }
    }
    if (row_half) {                                              // stored entries with j > i (what half_count_kernel finds)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) half += __shfl_xor(half, o, 64);
    }
    if (lane == 0) {
        cnt[i] = base < max_nbr ? base : max_nbr;
        if (row_half) row_half[i] = half;
        if (base > max_nbr) atomicMax(overflow, base);
    }
}

// ---------------------------------------------------------------------------- half list
__device__ __forceinline__ int first_greater(const int32_t* __restrict__ row, int n, int key) {
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (row[mid] > key) hi = mid; else lo = mid + 1; }
    return lo;
}

__global__ void half_count_kernel(const int32_t* __restrict__ col, const int32_t* __restrict__ cnt, int N,
                                  int max_nbr, int32_t* __restrict__ row_half, const int32_t* __restrict__ gate = nullptr) {
    if (gate && gate[0] == 0) return;                            // (Verlet reuse: the stored list still holds)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    row_half[i] = cnt[i] - first_greater(col + (size_t)i * max_nbr, cnt[i], i);
}

constexpr int HALF_FILL_WAVES = 4;
__global__ void half_fill_kernel(const int32_t* __restrict__ col, const int32_t* __restrict__ shift,
                                 const int32_t* __restrict__ cnt, const int32_t* __restrict__ row_base,
                                 int N, int max_nbr, int64_t* __restrict__ nbr, float* __restrict__ offsets,
                                 int32_t* __restrict__ edge_id, long long capacity, const int32_t* __restrict__ gate = nullptr,
                                 int pad = 0, float pad_offset = 0.f, int32_t* __restrict__ n_valid = nullptr,
                                 int32_t* __restrict__ need = nullptr) {
    if (gate && gate[0] == 0) return;                            // (Verlet reuse: the stored list still holds)
    if (pad) {                                                   // half_pad_kernel's sweep, spread over this grid
        const long long P = row_base[N];
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (n_valid) *n_valid = (int32_t)(P < capacity ? P : capacity);
            if (P > capacity && need) atomicMax(need, (int32_t)P);
        }
        for (long long e = P + (long long)blockIdx.x * blockDim.x + threadIdx.x; e < capacity; e += (long long)gridDim.x * blockDim.x) {
            nbr[2 * e] = -1; nbr[2 * e + 1] = -1;
            offsets[3 * e] = pad_offset; offsets[3 * e + 1] = 0.f; offsets[3 * e + 2] = 0.f;
        }
    }
    // one wave per atom, HALF_FILL_WAVES atoms per workgroup (a gated launch that finds nothing to do is N / 4 workgroup
    // starts instead of N: 7 -> 2.5 us at 32 768 atoms, 31 times per stacked SchNet pass)
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= N) return;
    const int32_t* row = col + (size_t)i * max_nbr;
    const int n = cnt[i];
    const int fg = first_greater(row, n, i);
    const int base = row_base[i];
    for (int k = threadIdx.x & 63; k < n; k += 64) {
        const int j = row[k];
        if (k >= fg) {
            const int e = base + (k - fg);
            if (e >= capacity) {                              // padded list too small (flagged by half_pad_kernel):
                if (edge_id) edge_id[(size_t)i * max_nbr + k] = 0;   // keep every edge id inside the buffers
                continue;
            }
            if (nbr) { nbr[2 * (size_t)e] = i; nbr[2 * (size_t)e + 1] = j; }
            if (offsets) {
                const int code = shift[(size_t)i * max_nbr + k];
                offsets[3 * (size_t)e] = (float)(code % 3 - 1);
                offsets[3 * (size_t)e + 1] = (float)((code / 3) % 3 - 1);
                offsets[3 * (size_t)e + 2] = (float)(code / 9 - 1);
            }
            if (edge_id) edge_id[(size_t)i * max_nbr + k] = e;
        } else if (edge_id) {
            // reverse slot: locate i inside row j
            const int32_t* rj = col + (size_t)j * max_nbr;
            const int nj = cnt[j];
            const int fgj = first_greater(rj, nj, j);
            const int pos_i = first_greater(rj, nj, i - 1);        // index of i in row j
            const int er = row_base[j] + (pos_i - fgj);
            edge_id[(size_t)i * max_nbr + k] = er < capacity ? er : 0;
        }
    }
}

// rows [P, capacity) of a fixed-capacity half list: sentinel pair (-1,-1) and an image flag that puts the
// "distance" far outside any radial basis; P > capacity is reported through need[0] (atomic max).
__global__ void half_pad_kernel(const int32_t* __restrict__ row_base, int N, long long capacity, float pad_offset,
                                int64_t* __restrict__ nbr, float* __restrict__ offsets, int32_t* __restrict__ n_valid,
                                int32_t* __restrict__ need, const int32_t* __restrict__ gate = nullptr) {
    if (gate && gate[0] == 0) return;                            // (Verlet reuse: the stored list still holds)
    const long long P = row_base[N];
    const long long e = P + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (n_valid) *n_valid = (int32_t)(P < capacity ? P : capacity);
        if (P > capacity && need) atomicMax(need, (int32_t)P);
    }
    if (e >= capacity) return;
    nbr[2 * e] = -1; nbr[2 * e + 1] = -1;
    offsets[3 * e] = pad_offset; offsets[3 * e + 1] = 0.f; offsets[3 * e + 2] = 0.f;
}

// ---------------------------------------------------------------------------- Verlet reuse of a stored list
// state[0] <- 1 when some atom is farther than `thr` from where the stored list was built (or the list has never been
// built: pos_build holds NaN), else 0; state[3] counts the builds.  state[1] / state[2] are the flag / ticket words of the
// last-block pattern (left zero for the next call).  No float atomics; the result does not depend on block order.
__global__ void verlet_check_kernel(const float* __restrict__ pos, const float* __restrict__ pos_build, int N, float thr2,
                                    int32_t* __restrict__ state) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool far = false;
    if (i < N) {
        const float dx = pos[3 * i] - pos_build[3 * i], dy = pos[3 * i + 1] - pos_build[3 * i + 1],
                    dz = pos[3 * i + 2] - pos_build[3 * i + 2];
        far = !(dx * dx + dy * dy + dz * dz <= thr2);            // (NaN compares false: a list never built is rebuilt)
    }
    const int any = __syncthreads_or(far ? 1 : 0);
    if (threadIdx.x != 0) return;
    if (any) atomicOr(&state[1], 1);
    __threadfence();
    const int ticket = atomicAdd(&state[2], 1);
    if (ticket == (int)gridDim.x - 1) {
        const int all = atomicOr(&state[1], 0);
        state[0] = all ? 1 : 0;
        if (all) state[3] += 1;
        state[1] = 0;
        state[2] = 0;
        __threadfence();
    }
}

__global__ void zero_i32_kernel(int32_t* __restrict__ p, int n, const int32_t* __restrict__ gate) {
    if (gate && gate[0] == 0) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}

}  // namespace

extern "C" int mdg_nbr_build_dense_groups(const float* pos, int n_atoms, int group, const MdgCell* cell,
                                          float cutoff, const uint8_t* mask, int32_t* col, int32_t* shift,
                                          int32_t* cnt, int max_nbr, int32_t* overflow, void* stream);

extern "C" int mdg_nbr_build_dense(const float* pos, int n_atoms, const MdgCell* cell, float cutoff,
                                   const uint8_t* mask, int32_t* col, int32_t* shift, int32_t* cnt,
                                   int max_nbr, int32_t* overflow, void* stream) {
    return mdg_nbr_build_dense_groups(pos, n_atoms, n_atoms, cell, cutoff, mask, col, shift, cnt, max_nbr, overflow,
                                      stream);
}

extern "C" int mdg_nbr_build_dense_groups(const float* pos, int n_atoms, int group, const MdgCell* cell,
                                          float cutoff, const uint8_t* mask, int32_t* col, int32_t* shift,
                                          int32_t* cnt, int max_nbr, int32_t* overflow, void* stream) {
    MDG_CHECK_ARG(pos && cell && col && shift && cnt && overflow, "nbr_build_dense: null buffer");
    MDG_CHECK_ARG(n_atoms > 0 && max_nbr > 0 && cutoff > 0.f, "nbr_build_dense: bad sizes");
    MDG_CHECK_ARG(group > 0 && n_atoms % group == 0, "nbr_build_dense: n_atoms must be a multiple of the group size");
    hipStream_t st = (hipStream_t)stream;
    const int wpb = 4;
    dim3 grid((n_atoms + wpb - 1) / wpb), block(64 * wpb);
    const float rc2 = cutoff * cutoff;
    if (cell->diag)
        hipLaunchKernelGGL(nbr_dense_kernel<true>, grid, block, 0, st, pos, n_atoms, group, *cell, rc2, mask, col,
                           shift, cnt, max_nbr, overflow);
    else
        hipLaunchKernelGGL(nbr_dense_kernel<false>, grid, block, 0, st, pos, n_atoms, group, *cell, rc2, mask, col,
                           shift, cnt, max_nbr, overflow);
    MDG_CHECK_LAUNCH("nbr_dense_kernel");
    return MDG_OK;
}

extern "C" int64_t mdg_nbr_cell_scratch_groups(int n_atoms, int group, const MdgCell* cell, float cutoff) {
    if (!cell || n_atoms <= 0 || group <= 0 || n_atoms % group || cutoff <= 0.f) return -1;
    const Bins b = make_bins(*cell, cutoff);
    return 2 * (int64_t)n_atoms + 3 * ((int64_t)b.ncell * (n_atoms / group) + 1) + 8;
}

extern "C" int64_t mdg_nbr_cell_scratch(int n_atoms, const MdgCell* cell, float cutoff) {
    return mdg_nbr_cell_scratch_groups(n_atoms, n_atoms, cell, cutoff);
}

extern "C" int mdg_nbr_build_cell_groups(const float* pos, int n_atoms, int group, const MdgCell* cell, float cutoff,
                                         const uint8_t* mask, int32_t* col, int32_t* shift, int32_t* cnt,
                                         int max_nbr, int32_t* overflow, int32_t* scratch, void* stream);

extern "C" int mdg_nbr_build_cell(const float* pos, int n_atoms, const MdgCell* cell, float cutoff,
                                  const uint8_t* mask, int32_t* col, int32_t* shift, int32_t* cnt,
                                  int max_nbr, int32_t* overflow, int32_t* scratch, void* stream) {
    return mdg_nbr_build_cell_groups(pos, n_atoms, n_atoms, cell, cutoff, mask, col, shift, cnt, max_nbr, overflow,
                                     scratch, stream);
}

extern "C" int mdg_nbr_build_cell_groups(const float* pos, int n_atoms, int group, const MdgCell* cell, float cutoff,
                                         const uint8_t* mask, int32_t* col, int32_t* shift, int32_t* cnt,
                                         int max_nbr, int32_t* overflow, int32_t* scratch, void* stream) {
    MDG_CHECK_ARG(pos && cell && col && shift && cnt && overflow && scratch, "nbr_build_cell: null buffer");
    MDG_CHECK_ARG(n_atoms > 0 && max_nbr > 0 && cutoff > 0.f, "nbr_build_cell: bad sizes");
    MDG_CHECK_ARG(group > 0 && n_atoms % group == 0, "nbr_build_cell: n_atoms must be a multiple of the group size");
    MDG_CHECK_ARG(cell->diag, "nbr_build_cell: orthorhombic cells only (use mdg_nbr_build_dense)");
    Bins bins = make_bins(*cell, cutoff);
    const int nbins_total = bins.ncell * (n_atoms / group);
    MDG_CHECK_ARG(bins.nb[0] >= 3 && bins.nb[1] >= 3 && bins.nb[2] >= 3,
                  "nbr_build_cell: box shorter than 3 cutoffs (use mdg_nbr_build_dense)");
    MDG_CHECK_ARG(max_nbr <= ROW_CAP, "nbr_build_cell: max_nbr > %d", ROW_CAP);
    hipStream_t st = (hipStream_t)stream;
    int32_t* atom_bin = scratch;
    int32_t* sorted_atoms = atom_bin + n_atoms;
    int32_t* bin_cnt = sorted_atoms + n_atoms;
    int32_t* bin_start = bin_cnt + nbins_total + 1;
    int32_t* cursor = bin_start + nbins_total + 1;
    if (hipMemsetAsync(bin_cnt, 0, sizeof(int32_t) * (nbins_total + 1), st) != hipSuccess) {
        mdg_set_error("nbr_build_cell: memset failed"); return MDG_ELAUNCH;
    }
    const int tb = 256;
    hipLaunchKernelGGL(bin_count_kernel, dim3((n_atoms + tb - 1) / tb), dim3(tb), 0, st, pos, n_atoms, group, *cell,
                       bins, atom_bin, bin_cnt);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, bin_cnt, nbins_total, bin_start, cursor);
    hipLaunchKernelGGL(bin_fill_kernel, dim3((n_atoms + tb - 1) / tb), dim3(tb), 0, st, atom_bin, n_atoms,
                       cursor, sorted_atoms);
    const int wpb = 4;
    const size_t lds = sizeof(int32_t) * 2 * ROW_CAP * wpb;
    hipLaunchKernelGGL(nbr_cell_kernel, dim3((n_atoms + wpb - 1) / wpb), dim3(64 * wpb), lds, st, pos, n_atoms, group,
                       *cell, bins, cutoff * cutoff, mask, atom_bin, bin_start, sorted_atoms, col, shift, cnt,
                       max_nbr, overflow);
    MDG_CHECK_LAUNCH("nbr_cell kernels");
    return MDG_OK;
}

extern "C" int mdg_nbr_half_count(const int32_t* col, const int32_t* cnt, int n_atoms, int max_nbr,
                                  int32_t* row_base, void* stream) {
    MDG_CHECK_ARG(col && cnt && row_base && n_atoms > 0, "nbr_half_count: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    // row_base doubles as the per-row count buffer before the in-place scan
    hipLaunchKernelGGL(half_count_kernel, dim3((n_atoms + 255) / 256), dim3(256), 0, st, col, cnt, n_atoms,
                       max_nbr, row_base);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, row_base, n_atoms, row_base, (int32_t*)nullptr);
    MDG_CHECK_LAUNCH("nbr_half_count");
    return MDG_OK;
}

extern "C" int mdg_nbr_half_fill(const int32_t* col, const int32_t* shift, const int32_t* cnt,
                                 const int32_t* row_base, int n_atoms, int max_nbr, int64_t* nbr,
                                 float* offsets, int32_t* edge_id, void* stream) {
    MDG_CHECK_ARG(col && shift && cnt && row_base && n_atoms > 0, "nbr_half_fill: bad arguments");
    hipLaunchKernelGGL(half_fill_kernel, dim3((n_atoms + HALF_FILL_WAVES - 1) / HALF_FILL_WAVES), dim3(64 * HALF_FILL_WAVES), 0, (hipStream_t)stream, col, shift, cnt,
                       row_base, n_atoms, max_nbr, nbr, offsets, edge_id, (long long)1 << 62);
    MDG_CHECK_LAUNCH("nbr_half_fill");
    return MDG_OK;
}

extern "C" int mdg_nbr_half_fill_padded(const int32_t* col, const int32_t* shift, const int32_t* cnt,
                                        const int32_t* row_base, int n_atoms, int max_nbr, int64_t capacity,
                                        float pad_offset, int64_t* nbr, float* offsets, int32_t* edge_id,
                                        int32_t* n_valid, int32_t* need, void* stream) {
    MDG_CHECK_ARG(col && shift && cnt && row_base && nbr && offsets && n_atoms > 0 && capacity > 0,
                  "nbr_half_fill_padded: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(half_fill_kernel, dim3((n_atoms + HALF_FILL_WAVES - 1) / HALF_FILL_WAVES), dim3(64 * HALF_FILL_WAVES), 0, st, col, shift, cnt, row_base, n_atoms,
                       max_nbr, nbr, offsets, edge_id, (long long)capacity);
    // the pad sweep covers the whole capacity (the pair count is only known on the device)
    hipLaunchKernelGGL(half_pad_kernel, dim3((unsigned)((capacity + 255) / 256)), dim3(256), 0, st, row_base,
                       n_atoms, (long long)capacity, pad_offset, nbr, offsets, n_valid, need);
    MDG_CHECK_LAUNCH("nbr_half_fill_padded");
    return MDG_OK;
}

// Fixed-capacity list (per-atom rows + padded half list) with Verlet reuse, one call per force evaluation and no host
// synchronisation: the list is searched with `list_cutoff` = cutoff + skin and kept while every atom stays within
// `half_skin` of where it was built (verlet_check_kernel decides on the device; the builder launches below return at
// their first instruction otherwise -- a captured HIP graph replays the same nodes either way).  Consumers re-apply the
// exact cutoff per pair (mdg_edge_geom_masked, mdg_pair_eval_ell_into's recheck bit), so the pair set of every evaluation
// is the one a fresh search at `cutoff` finds.  state: int32[4] (zero-initialised once; state[3] counts the builds),
// pos_build [N,3] initialised with NaN.  The other buffers are the ones mdg_nbr_build_*_groups / mdg_nbr_half_count /
// mdg_nbr_half_fill_padded take, all persistent.
extern "C" int mdg_nbr_verlet_rebuild(const float* pos, int n_atoms, int group, const MdgCell* cell, float list_cutoff,
                                      float half_skin, const uint8_t* mask, int use_cell_list, int32_t* col, int32_t* shift,
                                      int32_t* cnt, int max_nbr, int64_t capacity, float pad_offset, int64_t* nbr,
                                      float* offsets, int32_t* edge_id, int32_t* n_valid, int32_t* need, float* pos_build,
                                      int32_t* state, int32_t* row_base, int32_t* scratch, void* stream) {
    MDG_CHECK_ARG(pos && cell && col && shift && cnt && nbr && offsets && edge_id && n_valid && need && pos_build && state &&
                  row_base, "nbr_verlet_rebuild: null buffer");
    MDG_CHECK_ARG(n_atoms > 0 && max_nbr > 0 && list_cutoff > 0.f && half_skin >= 0.f && capacity > 0, "nbr_verlet_rebuild: bad sizes");
    MDG_CHECK_ARG(group > 0 && n_atoms % group == 0, "nbr_verlet_rebuild: n_atoms must be a multiple of the group size");
    hipStream_t st = (hipStream_t)stream;
    const int tb = 256;
    hipLaunchKernelGGL(verlet_check_kernel, dim3((n_atoms + tb - 1) / tb), dim3(tb), 0, st, pos, (const float*)pos_build, n_atoms,
                       half_skin * half_skin, state);
    const int32_t* gate = state;
    const float rc2 = list_cutoff * list_cutoff;
    if (use_cell_list) {
        MDG_CHECK_ARG(scratch, "nbr_verlet_rebuild: the cell-list search needs its scratch");
        MDG_CHECK_ARG(cell->diag, "nbr_verlet_rebuild: cell lists take orthorhombic cells only");
        Bins bins = make_bins(*cell, list_cutoff);
        const int nbins_total = bins.ncell * (n_atoms / group);
        MDG_CHECK_ARG(bins.nb[0] >= 3 && bins.nb[1] >= 3 && bins.nb[2] >= 3, "nbr_verlet_rebuild: box shorter than 3 list cutoffs");
        MDG_CHECK_ARG(max_nbr <= ROW_CAP, "nbr_verlet_rebuild: max_nbr > %d", ROW_CAP);
        int32_t* atom_bin = scratch;
        int32_t* sorted_atoms = atom_bin + n_atoms;
        int32_t* bin_cnt = sorted_atoms + n_atoms;
        int32_t* bin_start = bin_cnt + nbins_total + 1;
        int32_t* cursor = bin_start + nbins_total + 1;
        if (bins.ncell <= LDS_BINS && group <= SORT_MAX_GROUP) {
            hipLaunchKernelGGL(bin_sort_group_kernel, dim3(n_atoms / group), dim3(SORT_BLOCK), sizeof(int32_t) * 2 * bins.ncell, st,
                               pos, n_atoms, group, *cell, bins, atom_bin, bin_start, sorted_atoms, gate, pos_build);
        } else {
            hipLaunchKernelGGL(zero_i32_kernel, dim3((nbins_total + 1 + tb - 1) / tb), dim3(tb), 0, st, bin_cnt, nbins_total + 1, gate);
            hipLaunchKernelGGL(bin_count_kernel, dim3((n_atoms + tb - 1) / tb), dim3(tb), 0, st, pos, n_atoms, group, *cell, bins,
                               atom_bin, bin_cnt, gate, pos_build);
            hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, (const int32_t*)bin_cnt, nbins_total, bin_start, cursor, gate);
            hipLaunchKernelGGL(bin_fill_kernel, dim3((n_atoms + tb - 1) / tb), dim3(tb), 0, st, (const int32_t*)atom_bin, n_atoms,
                               cursor, sorted_atoms, gate);
        }
        const int wpb = 4;
        const size_t lds = sizeof(int32_t) * 2 * ROW_CAP * wpb;
        hipLaunchKernelGGL(nbr_cell_kernel, dim3((n_atoms + wpb - 1) / wpb), dim3(64 * wpb), lds, st, pos, n_atoms, group, *cell,
                           bins, rc2, mask, (const int32_t*)atom_bin, (const int32_t*)bin_start, (const int32_t*)sorted_atoms, col,
                           shift, cnt, max_nbr, need, gate, row_base);
    } else {
        const int wpb = 4;
        dim3 grid((n_atoms + wpb - 1) / wpb), block(64 * wpb);
        if (cell->diag)
            hipLaunchKernelGGL(nbr_dense_kernel<true>, grid, block, 0, st, pos, n_atoms, group, *cell, rc2, mask, col, shift, cnt,
                               max_nbr, need, gate, pos_build, row_base);
        else
            hipLaunchKernelGGL(nbr_dense_kernel<false>, grid, block, 0, st, pos, n_atoms, group, *cell, rc2, mask, col, shift, cnt,
                               max_nbr, need, gate, pos_build, row_base);
    }
    // (the searches leave the per-row half counts in row_base; the fill launch also pads [P, capacity))
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, (const int32_t*)row_base, n_atoms, row_base, (int32_t*)nullptr, gate);
    hipLaunchKernelGGL(half_fill_kernel, dim3((n_atoms + HALF_FILL_WAVES - 1) / HALF_FILL_WAVES), dim3(64 * HALF_FILL_WAVES), 0, st, (const int32_t*)col, (const int32_t*)shift,
                       (const int32_t*)cnt, (const int32_t*)row_base, n_atoms, max_nbr, nbr, offsets, edge_id, (long long)capacity, gate,
                       1, pad_offset, n_valid, need + 1);
    MDG_CHECK_LAUNCH("nbr_verlet_rebuild kernels");
    return MDG_OK;
}
