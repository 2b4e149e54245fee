// K8: soft-histogram radial distribution function and its gradient.
// Replaces rdf.forward (torchmd/observable.py:62-76: generate_nbr_list with cutoff end+0.5
// over all frames -> GaussianSmearing nff/nn/layers.py:14-31 -> sum) and the autograd
// backward through it.  The [pairs, bins] exp matrix of the reference is never
// materialised.
//
// forward : persistent blocks stride over (frame, 4096-candidate chunk) work items.  Threads
//           compute i<j minimum-image distances 256 candidates at a time, compact the accepted
//           ones into LDS in a fixed order, then thread k owns bin k and sweeps the LDS
//           distances (broadcast reads, register accumulator kept across work items).  Per-block
//           partial histograms are added by a second kernel in a fixed order => no atomics,
//           reproducible.  exp uses the hardware v_exp_f32 path (__expf, rel. err ~1e-6).
// backward: LPA lanes per (frame, atom); each accepted pair contributes
//           sum_k g_raw[k] * 2 coeff (d - mu_k) e_k  along the unit separation vector.
#include "common.hpp"

namespace {

constexpr int RDF_BLOCK = 256;
constexpr int RDF_MAX_BLOCKS = 2048;     // persistent blocks: each strides over (frame, chunk) work items
constexpr int RDF_CHUNK = 4096;          // candidate pairs per work item

// flat index c in [0, N(N-1)/2) -> (i, j), i < j, row-major (the order torch.nonzero yields)
__device__ __forceinline__ void pair_from_flat(long long c, int N, int& i, int& j) {
    const double b = 2.0 * N - 1.0;
    int ii = (int)((b - sqrt(b * b - 8.0 * (double)c)) * 0.5);
    // fix rounding: row ii starts at ii*(2N-ii-1)/2
    while ((long long)ii * (2LL * N - ii - 1) / 2 > c) --ii;
    while ((long long)(ii + 1) * (2LL * N - ii - 2) / 2 <= c) ++ii;
    i = ii;
    j = (int)(c - (long long)ii * (2LL * N - ii - 1) / 2) + ii + 1;
}

template <bool DIAG>
__global__ __launch_bounds__(RDF_BLOCK) void rdf_fwd_kernel(
    const float* __restrict__ xyz, int nF, int N, MdgCell cell, float rc2, const uint8_t* __restrict__ mask,
    const float* __restrict__ mu, float coeff, int nbins, float* __restrict__ partial) {
    __shared__ float dist[RDF_BLOCK];
    __shared__ int wcnt[RDF_BLOCK / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    constexpr int nw = RDF_BLOCK / 64;
    // register accumulators.  nbins >= RDF_BLOCK: thread owns bins k0 + m*RDF_BLOCK, m < 4, and
    // sweeps every distance.  nbins < RDF_BLOCK: G = RDF_BLOCK / nbins thread groups share the
    // sweep (group g takes distances g, g+G, ...) and are combined in group order at the end.
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const int G = nbins < RDF_BLOCK ? RDF_BLOCK / nbins : 1;
    const int grp = nbins < RDF_BLOCK ? threadIdx.x / nbins : 0;
    const int k0 = nbins < RDF_BLOCK ? (grp < G ? threadIdx.x % nbins : nbins) : threadIdx.x;
    const int k1 = k0 + RDF_BLOCK, k2 = k1 + RDF_BLOCK, k3 = k2 + RDF_BLOCK;
    const float m0 = k0 < nbins ? mu[k0] : 0.f, m1 = k1 < nbins ? mu[k1] : 0.f,
                m2 = k2 < nbins ? mu[k2] : 0.f, m3 = k3 < nbins ? mu[k3] : 0.f;
    const long long npair = (long long)N * (N - 1) / 2;
    const int chunks = (int)((npair + RDF_CHUNK - 1) / RDF_CHUNK);
    const long long items = (long long)nF * chunks;
    for (long long it = blockIdx.x; it < items; it += gridDim.x) {
        const int fr = (int)(it / chunks), ch = (int)(it % chunks);
        const float* pos = xyz + (size_t)fr * N * 3;
        const long long c_end = min(npair, (long long)(ch + 1) * RDF_CHUNK);
        for (long long cb = (long long)ch * RDF_CHUNK; cb < c_end; cb += RDF_BLOCK) {
            const long long c = cb + threadIdx.x;
            float d = -1.f;
            if (c < c_end) {
                int i, j;
                pair_from_flat(c, N, i, j);
                float dx = pos[3 * j] - pos[3 * i], dy = pos[3 * j + 1] - pos[3 * i + 1],
                      dz = pos[3 * j + 2] - pos[3 * i + 2];
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
#pragma unroll
            for (int w = 0; w < nw; ++w) { if (w < wid) base += wcnt[w]; total += wcnt[w]; }
            if (d >= 0.f) dist[base + __popcll(b & ((1ull << lane) - 1ull))] = d;
            __syncthreads();
            if (k0 < nbins) {
                for (int p = grp; p < total; p += G) {
                    const float dd = dist[p];
                    { const float x = dd - m0; a0 += __expf(coeff * x * x); }
                    if (k1 < nbins) { const float x = dd - m1; a1 += __expf(coeff * x * x); }
                    if (k2 < nbins) { const float x = dd - m2; a2 += __expf(coeff * x * x); }
                    if (k3 < nbins) { const float x = dd - m3; a3 += __expf(coeff * x * x); }
                }
            }
        }
    }
    float* out = partial + (size_t)blockIdx.x * nbins;
    if (G > 1) {
        __syncthreads();
        if (k0 < nbins) dist[grp * nbins + k0] = a0;           // G * nbins <= RDF_BLOCK
        __syncthreads();
        if (threadIdx.x < nbins) {
            float s = 0.f;
            for (int g = 0; g < G; ++g) s += dist[g * nbins + threadIdx.x];
            out[threadIdx.x] = s;
        }
        return;
    }
    if (k0 < nbins) out[k0] = a0;
    if (k1 < nbins) out[k1] = a1;
    if (k2 < nbins) out[k2] = a2;
    if (k3 < nbins) out[k3] = a3;
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
            s += sg[k] * (2.f * coeff * x) * __expf(coeff * x * x);
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

static int rdf_grid(int n_frames, int n_atoms) {
    const long long npair = (long long)n_atoms * (n_atoms - 1) / 2;
    const long long items = (long long)n_frames * ((npair + RDF_CHUNK - 1) / RDF_CHUNK);
    return (int)(items < RDF_MAX_BLOCKS ? items : RDF_MAX_BLOCKS);
}

extern "C" int64_t mdg_rdf_partial_size(int n_frames, int n_atoms, int nbins) {
    return (int64_t)rdf_grid(n_frames, n_atoms) * nbins;
}

extern "C" int mdg_rdf_fwd(const float* xyz, int n_frames, int n_atoms, const MdgCell* cell, float cutoff,
                           const uint8_t* mask, const float* mu, float coeff, int nbins, float* raw,
                           float* partial, void* stream) {
    MDG_CHECK_ARG(xyz && cell && mu && raw && partial, "rdf_fwd: null buffer");
    MDG_CHECK_ARG(n_frames > 0 && n_atoms > 1 && nbins > 0, "rdf_fwd: bad sizes");
    MDG_CHECK_ARG(nbins <= 4 * RDF_BLOCK, "rdf_fwd: nbins > %d not supported", 4 * RDF_BLOCK);
    const int nblocks = rdf_grid(n_frames, n_atoms);
    hipStream_t st = (hipStream_t)stream;
    if (cell->diag)
        hipLaunchKernelGGL(rdf_fwd_kernel<true>, dim3(nblocks), dim3(RDF_BLOCK), 0, st, xyz, n_frames, n_atoms,
                           *cell, cutoff * cutoff, mask, mu, coeff, nbins, partial);
    else
        hipLaunchKernelGGL(rdf_fwd_kernel<false>, dim3(nblocks), dim3(RDF_BLOCK), 0, st, xyz, n_frames, n_atoms,
                           *cell, cutoff * cutoff, mask, mu, coeff, nbins, partial);
    hipLaunchKernelGGL(rdf_finish_kernel, dim3(nbins), dim3(64), 0, st, partial, nblocks, nbins, raw);
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
