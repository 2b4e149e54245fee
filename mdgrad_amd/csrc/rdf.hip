// K8: soft-histogram radial distribution function and its gradient.
// Replaces rdf.forward (torchmd/observable.py:62-76: generate_nbr_list with cutoff end+0.5
// over all frames -> GaussianSmearing nff/nn/layers.py:14-31 -> sum) and the autograd
// backward through it.  The [pairs, bins] exp matrix of the reference is never
// materialised.
//
// forward : lane = pair; persistent waves stride over (frame, 1024-pair chunk) work items and
//           add each pair's windowed Gaussians into a wave-private LDS histogram (see
//           rdf_fwd_kernel).
// backward: one wave per frame in round-robin-tournament pair order (see rdf_bwd_kernel); each
//           accepted pair contributes sum_k g_raw[k] * 2 coeff (d - mu_k) e_k along +/- its unit
//           separation vector, accumulated in LDS without atomics.
#include "common.hpp"

namespace {

constexpr int RDF_BLOCK = 256;
constexpr int RDF_MAX_BLOCKS = 4096;     // persistent blocks: each strides over (frame, chunk) work items
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

// Forward: lane = pair.  exp(coeff (d - mu_k)^2) = exp2(-x_k^2), x_k = s (d - mu_k),
// s = sqrt(-coeff log2 e).  For equally spaced centres (mu = linspace, spacing D, Ds = s D) the
// Gaussians of one distance obey a two-term recurrence away from the nearest centre kc:
//     e_{m+1} = e_m rho_m ,  rho_{m+1} = rho_m exp2(-2 Ds^2) ,  rho_0 = exp2(+-2 Ds x_c - Ds^2)
// so a pair costs 3 v_exp_f32 plus two multiplies per bin, and only the bins within
// `reach` = 11.3/s of the distance are touched (beyond that the Gaussian is < 2^-126).  The
// values are accumulated into a wave-private LDS histogram with ds_add_f32 (no cross-wave
// traffic; within a wave the order is program order), the four wave histograms of a workgroup
// are combined in a fixed order, and a second kernel adds the per-workgroup partials in order.
// Error of the recurrence grows like m^2 ulp away from the centre, i.e. it is largest (~5e-6
// relative) where the Gaussian itself is ~1e-37.
constexpr int RDF_PAIRS_PER_ITEM = 1024;

template <bool DIAG>
__global__ __launch_bounds__(RDF_BLOCK) void rdf_fwd_kernel(
    const float* __restrict__ xyz, int nF, int N, MdgCell cell, float rc2, const uint8_t* __restrict__ mask,
    const float* __restrict__ mu, float coeff, int nbins, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int nw = RDF_BLOCK / 64;
    float* smu = sm;                                   // [nbins]
    float* hist = sm + nbins;                          // [nw][nbins]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < nbins; k += RDF_BLOCK) smu[k] = mu[k];
    for (int k = threadIdx.x; k < nw * nbins; k += RDF_BLOCK) hist[k] = 0.f;
    __syncthreads();
    float* myh = hist + wid * nbins;
    const float sc = sqrtf(-coeff * LOG2E);
    const float mu0 = smu[0];
    const float dmu = nbins > 1 ? (smu[nbins - 1] - mu0) / (float)(nbins - 1) : 1.f;
    const float inv_dmu = 1.0f / dmu;
    const float Ds = dmu * sc;
    const float c2 = __builtin_amdgcn_exp2f(-2.f * Ds * Ds);
    const int W = (int)ceilf(11.3f / Ds) + 1;          // bins on each side of the nearest centre
    const long long npair = (long long)N * (N - 1) / 2;
    const int chunks = (int)((npair + RDF_PAIRS_PER_ITEM - 1) / RDF_PAIRS_PER_ITEM);
    const long long items = (long long)nF * chunks;
    const long long gw = (long long)blockIdx.x * nw + wid, gstride = (long long)gridDim.x * nw;
    for (long long it = gw; it < items; it += gstride) {
        const int fr = (int)(it / chunks), ch = (int)(it % chunks);
        const float* pos = xyz + (size_t)fr * N * 3;
        const long long c_end = min(npair, (long long)(ch + 1) * RDF_PAIRS_PER_ITEM);
        long long c = (long long)ch * RDF_PAIRS_PER_ITEM + lane;
        int i = 0, j = 0;
        if (c < c_end) pair_from_flat(c, N, i, j);
        for (; c < c_end; c += 64) {
            float dx = pos[3 * j] - pos[3 * i], dy = pos[3 * j + 1] - pos[3 * i + 1], dz = pos[3 * j + 2] - pos[3 * i + 2];
            min_image<DIAG>(cell, dx, dy, dz);
            const float d2 = norm2_ref(dx, dy, dz);
            bool ok = (d2 < rc2) && (d2 != 0.f);
            if (ok && mask) ok = mask[(size_t)i * N + j] != 0;
            if (ok) {
                const float d = sqrtf(d2);
                const int kc = (int)rintf((d - mu0) * inv_dmu);
                if (kc >= -W && kc < nbins + W) {
                    const float muc = (kc >= 0 && kc < nbins) ? smu[kc] : mu0 + (float)kc * dmu;
                    const float xc = (d - muc) * sc;
                    const float ec = __builtin_amdgcn_exp2f(-xc * xc);
                    if (kc >= 0 && kc < nbins) atomicAdd(&myh[kc], ec);
                    float eu = ec, ed = ec;
                    float ru = __builtin_amdgcn_exp2f(2.f * Ds * xc - Ds * Ds);
                    float rd = __builtin_amdgcn_exp2f(-2.f * Ds * xc - Ds * Ds);
                    for (int m = 1; m <= W; ++m) {
                        eu *= ru; ru *= c2;
                        ed *= rd; rd *= c2;
                        const int ku = kc + m, kd = kc - m;
                        if (ku >= 0 && ku < nbins) atomicAdd(&myh[ku], eu);
                        if (kd >= 0 && kd < nbins) atomicAdd(&myh[kd], ed);
                    }
                }
            }
            j += 64;
            while (j >= N && i < N - 1) { ++i; j = j - N + i + 1; }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nbins; k += RDF_BLOCK) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < nw; ++w) s += hist[w * nbins + k];
        partial[(size_t)blockIdx.x * nbins + k] = s;
    }
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
template <bool DIAG>
__global__ __launch_bounds__(256) void rdf_bwd_kernel(
    const float* __restrict__ xyz, int nF, int N, MdgCell cell, float rc2, const uint8_t* __restrict__ mask,
    const float* __restrict__ mu, float coeff, int nbins, const float* __restrict__ g_raw,
    float* __restrict__ g_xyz) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* smu = sm;                         // [nbins]
    float* sg = sm + nbins;                  // [nbins]  g_k * 2 coeff / s
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float* px = sg + nbins + (size_t)wid * 6 * N;   // positions [3][N] then gradient [3][N] of this wave's frame
    float* gx = px + 3 * N;
    const float sc = sqrtf(-coeff * LOG2E);
    for (int k = threadIdx.x; k < nbins; k += blockDim.x) { smu[k] = mu[k]; sg[k] = g_raw[k] * 2.f * coeff / sc; }
    __syncthreads();
    const int fr = blockIdx.x * (blockDim.x >> 6) + wid;
    if (fr >= nF) return;
    const float* pos = xyz + (size_t)fr * N * 3;
    for (int e = lane; e < 3 * N; e += 64) { px[(e % 3) * N + e / 3] = pos[e]; gx[e] = 0.f; }
    const float dmu = nbins > 1 ? (smu[nbins - 1] - smu[0]) / (float)(nbins - 1) : 0.f;
    const float reach = 11.3f / sc;          // exp2 argument below -126 beyond this
    const float mu0 = smu[0];
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
                    int klo = 0, khi = nbins - 1;
                    if (dmu > 0.f) {
                        klo = max(0, (int)floorf((d - reach - mu0) / dmu));
                        khi = min(nbins - 1, (int)ceilf((d + reach - mu0) / dmu));
                    }
                    // dL/dd = sum_k g_k 2 coeff (d - mu_k) e_k = sum_k sg_k x_k exp2(-x_k^2), x_k = s (d - mu_k)
                    float sd = 0.f;
                    for (int k = klo; k <= khi; ++k) {
                        const float x = (d - smu[k]) * sc;
                        sd = fmaf(sg[k] * x, __builtin_amdgcn_exp2f(-x * x), sd);
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

}  // namespace

static int rdf_grid(int n_frames, int n_atoms) {
    const long long npair = (long long)n_atoms * (n_atoms - 1) / 2;
    const long long items = (long long)n_frames * ((npair + RDF_PAIRS_PER_ITEM - 1) / RDF_PAIRS_PER_ITEM);
    const long long blocks = (items + RDF_BLOCK / 64 - 1) / (RDF_BLOCK / 64);
    return (int)(blocks < RDF_MAX_BLOCKS ? blocks : RDF_MAX_BLOCKS);
}

extern "C" int64_t mdg_rdf_partial_size(int n_frames, int n_atoms, int nbins) {
    return (int64_t)rdf_grid(n_frames, n_atoms) * nbins;
}

extern "C" int mdg_rdf_fwd(const float* xyz, int n_frames, int n_atoms, const MdgCell* cell, float cutoff,
                           const uint8_t* mask, const float* mu, float coeff, int nbins, float* raw,
                           float* partial, void* stream) {
    MDG_CHECK_ARG(xyz && cell && mu && raw && partial, "rdf_fwd: null buffer");
    MDG_CHECK_ARG(n_frames > 0 && n_atoms > 1 && nbins > 0, "rdf_fwd: bad sizes");
    MDG_CHECK_ARG(nbins <= 8192, "rdf_fwd: nbins > 8192 not supported");
    MDG_CHECK_ARG(coeff < 0.f, "rdf_fwd: coeff must be negative (-0.5 / width^2)");
    const int nblocks = rdf_grid(n_frames, n_atoms);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = sizeof(float) * (size_t)nbins * (1 + RDF_BLOCK / 64);
    if (cell->diag)
        hipLaunchKernelGGL(rdf_fwd_kernel<true>, dim3(nblocks), dim3(RDF_BLOCK), lds, st, xyz, n_frames, n_atoms,
                           *cell, cutoff * cutoff, mask, mu, coeff, nbins, partial);
    else
        hipLaunchKernelGGL(rdf_fwd_kernel<false>, dim3(nblocks), dim3(RDF_BLOCK), lds, st, xyz, n_frames, n_atoms,
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
    MDG_CHECK_ARG(coeff < 0.f, "rdf_bwd: coeff must be negative (-0.5 / width^2)");
    hipStream_t st = (hipStream_t)stream;
    // waves (= frames) per workgroup limited by the 6N floats of LDS each one needs
    int wpb = 4;
    while (wpb > 1 && sizeof(float) * (2 * (size_t)nbins + (size_t)wpb * 6 * n_atoms) > 150 * 1024) wpb >>= 1;
    const size_t lds = sizeof(float) * (2 * (size_t)nbins + (size_t)wpb * 6 * n_atoms);
    MDG_CHECK_ARG(lds <= 160 * 1024, "rdf_bwd: N=%d does not fit the LDS-resident frame kernel", n_atoms);
    const int nblocks = (n_frames + wpb - 1) / wpb;
    if (cell->diag)
        hipLaunchKernelGGL(rdf_bwd_kernel<true>, dim3(nblocks), dim3(64 * wpb), lds, st, xyz, n_frames, n_atoms, *cell,
                           cutoff * cutoff, mask, mu, coeff, nbins, g_raw, g_xyz);
    else
        hipLaunchKernelGGL(rdf_bwd_kernel<false>, dim3(nblocks), dim3(64 * wpb), lds, st, xyz, n_frames, n_atoms, *cell,
                           cutoff * cutoff, mask, mu, coeff, nbins, g_raw, g_xyz);
    MDG_CHECK_LAUNCH("rdf_bwd_kernel");
    return MDG_OK;
}
