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
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
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
    const long long tot = (long long)n_edges * n_feat;
    hipLaunchKernelGGL(edge_prod_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b,
                       nbr, (long long)n_edges, n_feat, out);
    MDG_CHECK_LAUNCH("edge_prod_kernel");
    return MDG_OK;
}
