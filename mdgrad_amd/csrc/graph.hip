// K10 (+ its derivative closure): graph gather / scatter kernels for the SchNet continuous-
// filter convolution under GNNPotentials (nff/nn/graphconv.py:43-53, nff/nn/modules.py:564-571,
// nff/nn/models/schnet.py:142) on the per-atom (ELL) list with undirected edge ids.
//
//   edge_diff   out[e,c] = x[i_e,c] - x[j_e,c]                       (schnet.py:142 gathers)
//   edge_scatter out[n,c] = sum_{slots s of n} sgn_s g[eid_s,c]       (its transpose; sgn = +1 when n = i_e)
//   cfconv_agg  m[n,f]  = sum_{slots s of n} h[col_s,f] * W[eid_s,f] (message + both scatter_adds)
//   edge_prod   out[e,f] = a[i_e,f] b[j_e,f] + a[j_e,f] b[i_e,f]      (d cfconv_agg / dW)
//
// cfconv_agg is bilinear and symmetric in the adjacency, so {cfconv_agg, edge_prod} is closed
// under differentiation (d agg/dh = agg(., W), d agg/dW = edge_prod(h, .), d edge_prod/da =
// agg(b, .)); {edge_diff, edge_scatter} is a linear operator and its transpose.  All four are
// per-output gathers: no atomics, fixed summation order (rows are sorted by neighbour index).
// HBM-bound: feature rows are read with the feature index on the lanes (coalesced).
#include "common.hpp"

namespace {

__global__ void edge_diff_kernel(const float* __restrict__ x, const int64_t* __restrict__ nbr, long long E,
                                 int C, float* __restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * C) return;
    const long long e = t / C;
    const int c = (int)(t % C);
    const long long i = nbr[2 * e];
    // rows of a capacity-padded list (mdg_nbr_half_fill_padded) carry the sentinel -1: they yield 0
    out[t] = i < 0 ? 0.f : x[i * C + c] - x[nbr[2 * e + 1] * C + c];
}

// one wave per atom, lanes over the feature dimension (looped)
__global__ void edge_scatter_kernel(const float* __restrict__ g, const int32_t* __restrict__ col,
                                    const int32_t* __restrict__ eid, const int32_t* __restrict__ cnt,
                                    int N, int max_nbr, int C, float* __restrict__ out) {
    const int n = xcd_chunk(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    const int m = cnt[n];
    const size_t row = (size_t)n * max_nbr;
    for (int c = lane; c < C; c += 64) {
        float s = 0.f;
        for (int k = 0; k < m; ++k) {
            const float v = g[(size_t)eid[row + k] * C + c];
            s += col[row + k] > n ? v : -v;
        }
        out[(size_t)n * C + c] = s;
    }
}

__global__ void cfconv_agg_kernel(const float* __restrict__ h, const float* __restrict__ W,
                                  const int32_t* __restrict__ col, const int32_t* __restrict__ eid,
                                  const int32_t* __restrict__ cnt, int N, int max_nbr, int F,
                                  float* __restrict__ out) {
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    const int m = cnt[n];
    const size_t row = (size_t)n * max_nbr;
    for (int f = lane; f < F; f += 64) {
        float s = 0.f;
        int k = 0;
        for (; k + 1 < m; k += 2) {          // two independent gathers in flight
            const int j0 = col[row + k], j1 = col[row + k + 1];
            const int e0 = eid[row + k], e1 = eid[row + k + 1];
            const float a0 = h[(size_t)j0 * F + f] * W[(size_t)e0 * F + f];
            const float a1 = h[(size_t)j1 * F + f] * W[(size_t)e1 * F + f];
            s += a0;
            s += a1;
        }
        if (k < m) s += h[(size_t)col[row + k] * F + f] * W[(size_t)eid[row + k] * F + f];
        out[(size_t)n * F + f] = s;
    }
}

__global__ void edge_prod_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                 const int64_t* __restrict__ nbr, long long E, int F, float* __restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * F) return;
    const long long e = t / F;
    const int f = (int)(t % F);
    const long long i = nbr[2 * e], j = nbr[2 * e + 1];
    out[t] = i < 0 ? 0.f : a[i * F + f] * b[j * F + f] + a[j * F + f] * b[i * F + f];   // (-1: padding row)
}

// 16-byte variants for F = 4 * LPE with LPE (lanes per edge) a power of two <= 64: a wave works on
// 64 / LPE edges at a time; the per-group partial sums are combined in a fixed order.
template <int LPE>
__global__ void cfconv_agg_v4_kernel(const float4* __restrict__ h, const float4* __restrict__ W,
                                     const int32_t* __restrict__ col, const int32_t* __restrict__ eid,
                                     const int32_t* __restrict__ cnt, int N, int max_nbr, float4* __restrict__ out) {
    constexpr int G = 64 / LPE;
    const int n = xcd_chunk(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    const int grp = lane / LPE, f = lane % LPE;
    const int m = cnt[n];
    const size_t row = (size_t)n * max_nbr;
    float4 s = {0.f, 0.f, 0.f, 0.f};
    int k = grp;
    for (; k + G < m; k += 2 * G) {                       // two independent gathers in flight
        const int j0 = col[row + k], j1 = col[row + k + G];
        const int e0 = eid[row + k], e1 = eid[row + k + G];
        const float4 a0 = h[(size_t)j0 * LPE + f], w0 = W[(size_t)e0 * LPE + f];
        const float4 a1 = h[(size_t)j1 * LPE + f], w1 = W[(size_t)e1 * LPE + f];
        s.x += a0.x * w0.x; s.y += a0.y * w0.y; s.z += a0.z * w0.z; s.w += a0.w * w0.w;
        s.x += a1.x * w1.x; s.y += a1.y * w1.y; s.z += a1.z * w1.z; s.w += a1.w * w1.w;
    }
    if (k < m) {
        const float4 a0 = h[(size_t)col[row + k] * LPE + f], w0 = W[(size_t)eid[row + k] * LPE + f];
        s.x += a0.x * w0.x; s.y += a0.y * w0.y; s.z += a0.z * w0.z; s.w += a0.w * w0.w;
    }
#pragma unroll
    for (int o = LPE; o < 64; o <<= 1) {
        s.x += __shfl_xor(s.x, o, 64); s.y += __shfl_xor(s.y, o, 64);
        s.z += __shfl_xor(s.z, o, 64); s.w += __shfl_xor(s.w, o, 64);
    }
    if (grp == 0) out[(size_t)n * LPE + f] = s;
}

template <int LPE>
__global__ void edge_prod_v4_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                    const int64_t* __restrict__ nbr, long long E, float4* __restrict__ out) {
    const long long e = (long long)xcd_chunk(blockIdx.x, gridDim.x) * (blockDim.x / LPE) + threadIdx.x / LPE;
    const int f = threadIdx.x % LPE;
    if (e >= E) return;
    const long long i = nbr[2 * e], j = nbr[2 * e + 1];
    float4 r = {0.f, 0.f, 0.f, 0.f};
    if (i >= 0) {                                                    // (-1: padding row)
        const float4 ai = a[i * LPE + f], bj = b[j * LPE + f], aj = a[j * LPE + f], bi = b[i * LPE + f];
        r.x = ai.x * bj.x + aj.x * bi.x; r.y = ai.y * bj.y + aj.y * bi.y;
        r.z = ai.z * bj.z + aj.z * bi.z; r.w = ai.w * bj.w + aj.w * bi.w;
    }
    out[e * LPE + f] = r;
}

__host__ inline bool vec4_ok(int F, const void* p0, const void* p1, const void* p2) {
    const int l = F / 4;
    return F % 4 == 0 && l >= 4 && l <= 64 && (l & (l - 1)) == 0 &&
           (((uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2) & 15) == 0;
}

}  // namespace

extern "C" int mdg_edge_diff(const float* x, const int64_t* nbr, int64_t n_edges, int n_feat, float* out,
                             void* stream) {
    MDG_CHECK_ARG(n_edges >= 0 && n_feat > 0, "edge_diff: bad sizes");
    if (n_edges == 0) return MDG_OK;
    MDG_CHECK_ARG(x && nbr && out, "edge_diff: null buffer");
    const long long tot = (long long)n_edges * n_feat;
    hipLaunchKernelGGL(edge_diff_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       nbr, (long long)n_edges, n_feat, out);
    MDG_CHECK_LAUNCH("edge_diff_kernel");
    return MDG_OK;
}

extern "C" int mdg_edge_scatter(const float* g, const int32_t* col, const int32_t* eid, const int32_t* cnt,
                                int n_atoms, int max_nbr, int n_feat, float* out, void* stream) {
    MDG_CHECK_ARG(col && eid && cnt && out && n_atoms > 0 && n_feat > 0, "edge_scatter: bad arguments");
    hipLaunchKernelGGL(edge_scatter_kernel, dim3((n_atoms + 3) / 4), dim3(256), 0, (hipStream_t)stream, g, col, eid,
                       cnt, n_atoms, max_nbr, n_feat, out);
    MDG_CHECK_LAUNCH("edge_scatter_kernel");
    return MDG_OK;
}

extern "C" int mdg_cfconv_agg(const float* h, const float* W, const int32_t* col, const int32_t* eid,
                              const int32_t* cnt, int n_atoms, int max_nbr, int n_feat, float* out, void* stream) {
    MDG_CHECK_ARG(h && col && eid && cnt && out && n_atoms > 0 && n_feat > 0, "cfconv_agg: bad arguments");
    if (vec4_ok(n_feat, h, W, out)) {
#define MDG_AGG(L)                                                                                               \
    hipLaunchKernelGGL(cfconv_agg_v4_kernel<L>, dim3((n_atoms + 3) / 4), dim3(256), 0, (hipStream_t)stream,      \
                       (const float4*)h, (const float4*)W, col, eid, cnt, n_atoms, max_nbr, (float4*)out)
        switch (n_feat / 4) {
        case 4: MDG_AGG(4); break;
        case 8: MDG_AGG(8); break;
        case 16: MDG_AGG(16); break;
        case 32: MDG_AGG(32); break;
        default: MDG_AGG(64); break;
        }
#undef MDG_AGG
        MDG_CHECK_LAUNCH("cfconv_agg_v4_kernel");
        return MDG_OK;
    }
    hipLaunchKernelGGL(cfconv_agg_kernel, dim3((n_atoms + 3) / 4), dim3(256), 0, (hipStream_t)stream, h, W, col, eid,
                       cnt, n_atoms, max_nbr, n_feat, out);
    MDG_CHECK_LAUNCH("cfconv_agg_kernel");
    return MDG_OK;
}

extern "C" int mdg_edge_prod(const float* a, const float* b, const int64_t* nbr, int64_t n_edges, int n_feat,
                             float* out, void* stream) {
    MDG_CHECK_ARG(n_edges >= 0 && n_feat > 0, "edge_prod: bad sizes");
    if (n_edges == 0) return MDG_OK;
    MDG_CHECK_ARG(a && b && nbr && out, "edge_prod: null buffer");
    if (vec4_ok(n_feat, a, b, out)) {
#define MDG_PROD(L)                                                                                         \
    hipLaunchKernelGGL(edge_prod_v4_kernel<L>, dim3((unsigned)((n_edges + 256 / L - 1) / (256 / L))), dim3(256), 0, \
                       (hipStream_t)stream, (const float4*)a, (const float4*)b, nbr, (long long)n_edges, (float4*)out)
        switch (n_feat / 4) {
        case 4: MDG_PROD(4); break;
        case 8: MDG_PROD(8); break;
        case 16: MDG_PROD(16); break;
        case 32: MDG_PROD(32); break;
        default: MDG_PROD(64); break;
        }
#undef MDG_PROD
        MDG_CHECK_LAUNCH("edge_prod_v4_kernel");
        return MDG_OK;
    }
    const long long tot = (long long)n_edges * n_feat;
    hipLaunchKernelGGL(edge_prod_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b,
                       nbr, (long long)n_edges, n_feat, out);
    MDG_CHECK_LAUNCH("edge_prod_kernel");
    return MDG_OK;
}
