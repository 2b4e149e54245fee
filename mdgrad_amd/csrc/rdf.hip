// K8: soft-histogram radial distribution function and its gradient.
// Replaces rdf.forward (torchmd/observable.py:62-76: generate_nbr_list with cutoff end+0.5
// over all frames -> GaussianSmearing nff/nn/layers.py:14-31 -> sum) and the autograd
// backward through it.  The [pairs, bins] exp matrix of the reference is never
// materialised.
//
// forward : block = (frame, tile of TI rows).  Threads compute the tile's i<j minimum-image
//           distances 256 candidates at a time, compact the accepted ones into LDS in a
//           fixed order, then thread k owns bin k and sweeps the LDS distances (broadcast
//           reads, register accumulator).  Per-block partial histograms are added by a
//           second kernel in block order => no atomics, reproducible.
// backward: LPA lanes per (frame, atom); each accepted pair contributes
//           sum_k g_raw[k] * 2 coeff (d - mu_k) e_k  along the unit separation vector.
#include "common.hpp"

namespace {

constexpr int RDF_TI = 8;        // rows per forward block
constexpr int RDF_BLOCK = 256;

template <bool DIAG>
__global__ void rdf_fwd_kernel(const float* __restrict__ xyz, int nF, int N, MdgCell cell, float rc2,
                               const uint8_t* __restrict__ mask, const float* __restrict__ mu, float coeff,
                               int nbins, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* dist = sm;                      // [RDF_BLOCK]
    int* wcnt = (int*)(sm + RDF_BLOCK);    // [8]
    float* acc = sm + RDF_BLOCK + 8;       // [nbins] (used when nbins > blockDim)
    const int tiles = (N + RDF_TI - 1) / RDF_TI;
    const int fr = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const float* pos = xyz + (size_t)fr * N * 3;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // register accumulators for up to 4 bins per thread
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const int k0 = threadIdx.x, k1 = k0 + blockDim.x, k2 = k1 + blockDim.x, k3 = k2 + blockDim.x;
    const float m0 = k0 < nbins ? mu[k0] : 0.f, m1 = k1 < nbins ? mu[k1] : 0.f,
                m2 = k2 < nbins ? mu[k2] : 0.f, m3 = k3 < nbins ? mu[k3] : 0.f;
    (void)acc;
    const int i_end = min(N, (tile + 1) * RDF_TI);
    for (int i = tile * RDF_TI; i < i_end; ++i) {
        const float xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
        for (int jb = i + 1; jb < N; jb += blockDim.x) {
            const int j = jb + threadIdx.x;
            float d = -1.f;
            if (j < N) {
                float dx = pos[3 * j] - xi, dy = pos[3 * j + 1] - yi, dz = pos[3 * j + 2] - zi;
                min_image<DIAG>(cell, dx, dy, dz);
                const float d2 = norm2_ref(dx, dy, dz);
                bool ok = (d2 < rc2) && (d2 != 0.f);
                if (ok && mask) ok = mask[(size_t)i * N + j] != 0;
                if (ok) d = sqrtf(d2);
            }
            const unsigned long long b = __ballot(d >= 0.f);
            __syncthreads();                                   // previous sweep done with dist[]
            if (lane == 0) wcnt[wid] = __popcll(b);
            __syncthreads();
            int base = 0, total = 0;
            for (int w = 0; w < nw; ++w) { if (w < wid) base += wcnt[w]; total += wcnt[w]; }
            if (d >= 0.f) dist[base + __popcll(b & ((1ull << lane) - 1ull))] = d;
            __syncthreads();
            for (int p = 0; p < total; ++p) {
                const float dd = dist[p];
                if (k0 < nbins) { const float x = dd - m0; a0 += expf(coeff * x * x); }
                if (k1 < nbins) { const float x = dd - m1; a1 += expf(coeff * x * x); }
                if (k2 < nbins) { const float x = dd - m2; a2 += expf(coeff * x * x); }
                if (k3 < nbins) { const float x = dd - m3; a3 += expf(coeff * x * x); }
            }
        }
    }
    float* out = partial + (size_t)blockIdx.x * nbins;
    if (k0 < nbins) out[k0] = a0;
    if (k1 < nbins) out[k1] = a1;
    if (k2 < nbins) out[k2] = a2;
    if (k3 < nbins) out[k3] = a3;
}

__global__ void rdf_finish_kernel(const float* __restrict__ partial, int nblocks, int nbins,
                                  float* __restrict__ raw) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nbins) return;
    float s = 0.f;
    for (int b = 0; b < nblocks; ++b) s += partial[(size_t)b * nbins + k];
    raw[k] = s;
}

template <bool DIAG, int LPA>
__global__ void rdf_bwd_kernel(const float* __restrict__ xyz, int nF, int N, MdgCell cell, float rc2,
                               const uint8_t* __restrict__ mask, const float* __restrict__ mu, float coeff,
                               int nbins, const float* __restrict__ g_raw, float* __restrict__ g_xyz) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* smu = sm;             // [nbins]
    float* sg = sm + nbins;      // [nbins]
    for (int k = threadIdx.x; k < nbins; k += blockDim.x) { smu[k] = mu[k]; sg[k] = g_raw[k]; }
    __syncthreads();
    const int apb = blockDim.x / LPA;
    const long long gi = (long long)blockIdx.x * apb + threadIdx.x / LPA;
    const int sub = threadIdx.x % LPA;
    if (gi >= (long long)nF * N) return;
    const int fr = (int)(gi / N), i = (int)(gi % N);
    const float* pos = xyz + (size_t)fr * N * 3;
    const float xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
    // bins whose Gaussian is non-negligible (exp argument > -88) around a distance
    const float dmu = nbins > 1 ? (smu[nbins - 1] - smu[0]) / (float)(nbins - 1) : 0.f;
    const float reach = sqrtf(88.f / fabsf(coeff));
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
        const float d = sqrtf(d2);
        int klo = 0, khi = nbins - 1;
        if (dmu > 0.f) {
            klo = max(0, (int)floorf((d - reach - smu[0]) / dmu));
            khi = min(nbins - 1, (int)ceilf((d + reach - smu[0]) / dmu));
        }
        float s = 0.f;
        for (int k = klo; k <= khi; ++k) {
            const float x = d - smu[k];
            s += sg[k] * (2.f * coeff * x) * expf(coeff * x * x);
        }
        // d(dist)/dx_i = -(D)/d for D = x_j - x_i (unflipped); with flip, D was negated
        const float c = (flip ? s : -s) / d;
        gx = fmaf(c, dx, gx); gy = fmaf(c, dy, gy); gz = fmaf(c, dz, gz);
    }
    gx = group_sum<LPA>(gx); gy = group_sum<LPA>(gy); gz = group_sum<LPA>(gz);
    if (sub == 0) {
        float* o = g_xyz + ((size_t)fr * N + i) * 3;
        o[0] = gx; o[1] = gy; o[2] = gz;
    }
}

}  // namespace

extern "C" int64_t mdg_rdf_partial_size(int n_frames, int n_atoms, int nbins) {
    const int64_t tiles = (n_atoms + RDF_TI - 1) / RDF_TI;
    return (int64_t)n_frames * tiles * nbins;
}

extern "C" int mdg_rdf_fwd(const float* xyz, int n_frames, int n_atoms, const MdgCell* cell, float cutoff,
                           const uint8_t* mask, const float* mu, float coeff, int nbins, float* raw,
                           float* partial, void* stream) {
    MDG_CHECK_ARG(xyz && cell && mu && raw && partial, "rdf_fwd: null buffer");
    MDG_CHECK_ARG(n_frames > 0 && n_atoms > 1 && nbins > 0, "rdf_fwd: bad sizes");
    MDG_CHECK_ARG(nbins <= 4 * RDF_BLOCK, "rdf_fwd: nbins > %d not supported", 4 * RDF_BLOCK);
    const int tiles = (n_atoms + RDF_TI - 1) / RDF_TI;
    const int nblocks = n_frames * tiles;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = sizeof(float) * (RDF_BLOCK + 8 + nbins);
    if (cell->diag)
        hipLaunchKernelGGL(rdf_fwd_kernel<true>, dim3(nblocks), dim3(RDF_BLOCK), lds, st, xyz, n_frames, n_atoms,
                           *cell, cutoff * cutoff, mask, mu, coeff, nbins, partial);
    else
        hipLaunchKernelGGL(rdf_fwd_kernel<false>, dim3(nblocks), dim3(RDF_BLOCK), lds, st, xyz, n_frames, n_atoms,
                           *cell, cutoff * cutoff, mask, mu, coeff, nbins, partial);
    hipLaunchKernelGGL(rdf_finish_kernel, dim3((nbins + 63) / 64), dim3(64), 0, st, partial, nblocks, nbins, raw);
    MDG_CHECK_LAUNCH("rdf_fwd_kernel");
    return MDG_OK;
}

extern "C" int mdg_rdf_bwd(const float* xyz, int n_frames, int n_atoms, const MdgCell* cell, float cutoff,
                           const uint8_t* mask, const float* mu, float coeff, int nbins, const float* g_raw,
                           float* g_xyz, void* stream) {
    MDG_CHECK_ARG(xyz && cell && mu && g_raw && g_xyz, "rdf_bwd: null buffer");
    MDG_CHECK_ARG(n_frames > 0 && n_atoms > 1 && nbins > 0, "rdf_bwd: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    constexpr int LPA = 16;
    const int apb = RDF_BLOCK / LPA;
    const long long rows = (long long)n_frames * n_atoms;
    const int nblocks = (int)((rows + apb - 1) / apb);
    const size_t lds = sizeof(float) * 2 * nbins;
    if (cell->diag)
        hipLaunchKernelGGL((rdf_bwd_kernel<true, LPA>), dim3(nblocks), dim3(RDF_BLOCK), lds, st, xyz, n_frames,
                           n_atoms, *cell, cutoff * cutoff, mask, mu, coeff, nbins, g_raw, g_xyz);
    else
        hipLaunchKernelGGL((rdf_bwd_kernel<false, LPA>), dim3(nblocks), dim3(RDF_BLOCK), lds, st, xyz, n_frames,
                           n_atoms, *cell, cutoff * cutoff, mask, mu, coeff, nbins, g_raw, g_xyz);
    MDG_CHECK_LAUNCH("rdf_bwd_kernel");
    return MDG_OK;
}
