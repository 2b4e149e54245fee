// Fused elementwise / row-reduction pieces of the hand-derived SchNet passes
// (mdgrad_amd/nn/analytic.py): each replaces a chain of 4-14 PyTorch elementwise ops on [E,G] /
// [N,A] tensors (Gaussian smearing nff/nn/layers.py:14-31, shifted softplus
// nff/nn/activations.py:5-11 and their first/second-order derivative algebra).  HBM-bound:
// one read of each input, one write of each output, feature index on the lanes.
#include "common.hpp"

namespace {

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float ssp_f(float x) {        // softplus(x) - ln2 (torch threshold 20)
    return (x > 20.f ? x : log1pf(expf(x))) - 0.69314718055994531f;
}

// g = exp(c_k (d - mu_k)^2), phi = 2 c_k (d - mu_k)
__global__ void smear_kernel(const float* __restrict__ d, const float* __restrict__ mu, const float* __restrict__ c,
                             long long E, int G, float* __restrict__ g, float* __restrict__ phi) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * G) return;
    const int k = (int)(t % G);
    const float x = d[t / G] - mu[k], ck = c[k];
    g[t] = expf(ck * x * x);
    phi[t] = 2.f * ck * x;
}

// s = ssp(a); optionally sa = sigmoid(a)
__global__ void ssp_kernel(const float* __restrict__ a, long long n, float* __restrict__ s, float* __restrict__ sa) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float x = a[t];
    s[t] = ssp_f(x);
    if (sa) sa[t] = sigmoidf(x);
}

// tangent of ssp: sd = sa * ad   and   gd = g * phi * dd[row]  (two tiny fused products)
__global__ void mul2_kernel(const float* __restrict__ x, const float* __restrict__ y, long long n, float* __restrict__ o) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) o[t] = x[t] * y[t];
}
__global__ void mul_row_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ r,
                               long long E, int G, float* __restrict__ o) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < E * G) o[t] = x[t] * y[t] * r[t / G];
}

// reverse of the (ssp, its tangent) pair:  xdb = sa * sdb ;  xb = sa (1 - sa) xd sdb + sa sb
__global__ void ssp_dual_bwd_kernel(const float* __restrict__ sa, const float* __restrict__ xd,
                                    const float* __restrict__ sdb, const float* __restrict__ sb, long long n,
                                    float* __restrict__ xdb, float* __restrict__ xb) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float s = sa[t], db = sdb[t];
    xdb[t] = s * db;
    xb[t] = s * (1.f - s) * xd[t] * db + s * sb[t];
}

// the same with the tangent stored as t_dot = sa * x_dot (what the fused Dense epilogue writes):
//   xdb = sa * sdb ;  xb = (1 - sa) t_dot sdb + sa sb
__global__ void ssp_dual_bwd_t_kernel(const float* __restrict__ sa, const float* __restrict__ td,
                                      const float* __restrict__ sdb, const float* __restrict__ sb, long long n,
                                      float* __restrict__ xdb, float* __restrict__ xb) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float s = sa[t], db = sdb[t];
    xdb[t] = s * db;
    xb[t] = (1.f - s) * td[t] * db + s * sb[t];
}

// head of the reverse sweeps through the readout  U = sum_i L2 . ssp(y_i) + l2  (nff/nn/modules.py:761-809):
//   ydb = sy * L2  (the adjoint of y_dot, which is also dU/dy) ;  yb = (1 - sy) (sy y_dot) L2  (the adjoint of y in U_dot)
// from sy = sigmoid(y) and sy * y_dot as the Dense epilogue leaves them; syd = NULL: first-order pass, ydb only
__global__ void readout_head_kernel(const float* __restrict__ sy, const float* __restrict__ syd, const float* __restrict__ L2,
                                    long long n, int cols, float* __restrict__ ydb, float* __restrict__ yb) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float s = sy[t], l = L2[t % cols];
    ydb[t] = s * l;
    if (syd) yb[t] = (1.f - s) * syd[t] * l;
}

// reverse of the smearing (and its tangent), reduced over the G Gaussians of each edge:
//   gb' = gb + gdb * phi * dd ;  d_b += sum_k gdb g 2c dd + gb' g phi ;  dd_b += sum_k gdb g phi
// gdb may be NULL (first-order pass): d_b += sum_k gb g phi.   16 lanes per edge row.
__global__ void smear_bwd_kernel(const float* __restrict__ gdb, const float* __restrict__ gb,
                                 const float* __restrict__ g, const float* __restrict__ phi,
                                 const float* __restrict__ dd, const float* __restrict__ c, long long E, int G,
                                 float* __restrict__ d_b, float* __restrict__ dd_b) {
    const long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    float s1 = 0.f, s2 = 0.f;
    if (row < E) {
        const float ddr = gdb ? dd[row] : 0.f;
        for (int k = sub; k < G; k += 16) {
            const long long t = row * G + k;
            const float gg = g[t], ph = phi[t];
            float b = gb[t];
            if (gdb) {
                const float db = gdb[t];
                b += db * ph * ddr;
                s1 += db * gg * 2.f * c[k] * ddr;
                s2 += db * gg * ph;
            }
            s1 += b * gg * ph;
        }
    }
    s1 = group_sum<16>(s1);
    s2 = group_sum<16>(s2);
    if (row < E && sub == 0) {
        d_b[row] += s1;
        if (gdb) dd_b[row] += s2;
    }
}

inline unsigned nblk(long long n, int b = 256) { return (unsigned)((n + b - 1) / b); }

}  // namespace

#define ELEM_CHECK(n, name)                                  \
    MDG_CHECK_ARG((n) >= 0, name ": bad size");              \
    if ((n) == 0) return MDG_OK;

extern "C" int mdg_smear(const float* d, const float* mu, const float* c, int64_t n_edges, int n_gauss, float* g,
                         float* phi, void* stream) {
    ELEM_CHECK(n_edges, "smear");
    MDG_CHECK_ARG(d && mu && c && g && phi && n_gauss > 0, "smear: bad arguments");
    hipLaunchKernelGGL(smear_kernel, dim3(nblk(n_edges * n_gauss)), dim3(256), 0, (hipStream_t)stream, d, mu, c,
                       (long long)n_edges, n_gauss, g, phi);
    MDG_CHECK_LAUNCH("smear_kernel");
    return MDG_OK;
}

extern "C" int mdg_ssp(const float* a, int64_t n, float* s, float* sa, void* stream) {
    ELEM_CHECK(n, "ssp");
    MDG_CHECK_ARG(a && s, "ssp: bad arguments");
    hipLaunchKernelGGL(ssp_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, a, (long long)n, s, sa);
    MDG_CHECK_LAUNCH("ssp_kernel");
    return MDG_OK;
}

extern "C" int mdg_mul_row(const float* x, const float* y, const float* r, int64_t n_rows, int n_cols, float* o,
                           void* stream) {
    ELEM_CHECK(n_rows, "mul_row");
    MDG_CHECK_ARG(x && y && o && n_cols > 0, "mul_row: bad arguments");
    if (r)
        hipLaunchKernelGGL(mul_row_kernel, dim3(nblk(n_rows * n_cols)), dim3(256), 0, (hipStream_t)stream, x, y, r,
                           (long long)n_rows, n_cols, o);
    else
        hipLaunchKernelGGL(mul2_kernel, dim3(nblk(n_rows * n_cols)), dim3(256), 0, (hipStream_t)stream, x, y,
                           (long long)n_rows * n_cols, o);
    MDG_CHECK_LAUNCH("mul_row_kernel");
    return MDG_OK;
}

extern "C" int mdg_ssp_dual_bwd(const float* sa, const float* xd, const float* sdb, const float* sb, int64_t n,
                                float* xdb, float* xb, void* stream) {
    ELEM_CHECK(n, "ssp_dual_bwd");
    MDG_CHECK_ARG(sa && xd && sdb && sb && xdb && xb, "ssp_dual_bwd: bad arguments");
    hipLaunchKernelGGL(ssp_dual_bwd_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, sa, xd, sdb, sb,
                       (long long)n, xdb, xb);
    MDG_CHECK_LAUNCH("ssp_dual_bwd_kernel");
    return MDG_OK;
}

extern "C" int mdg_ssp_dual_bwd_t(const float* sa, const float* td, const float* sdb, const float* sb, int64_t n,
                                  float* xdb, float* xb, void* stream) {
    ELEM_CHECK(n, "ssp_dual_bwd_t");
    MDG_CHECK_ARG(sa && td && sdb && sb && xdb && xb, "ssp_dual_bwd_t: bad arguments");
    hipLaunchKernelGGL(ssp_dual_bwd_t_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, sa, td, sdb, sb,
                       (long long)n, xdb, xb);
    MDG_CHECK_LAUNCH("ssp_dual_bwd_t_kernel");
    return MDG_OK;
}

extern "C" int mdg_readout_head(const float* sy, const float* syd, const float* L2, int64_t n_rows, int n_cols, float* ydb,
                                float* yb, void* stream) {
    ELEM_CHECK(n_rows, "readout_head");
    MDG_CHECK_ARG(sy && L2 && ydb && n_cols > 0 && (!syd || yb), "readout_head: bad arguments");
    hipLaunchKernelGGL(readout_head_kernel, dim3(nblk(n_rows * n_cols)), dim3(256), 0, (hipStream_t)stream, sy, syd, L2,
                       (long long)n_rows * n_cols, n_cols, ydb, yb);
    MDG_CHECK_LAUNCH("readout_head_kernel");
    return MDG_OK;
}

extern "C" int mdg_smear_bwd(const float* gdb, const float* gb, const float* g, const float* phi, const float* dd,
                             const float* c, int64_t n_edges, int n_gauss, float* d_b, float* dd_b, void* stream) {
    ELEM_CHECK(n_edges, "smear_bwd");
    MDG_CHECK_ARG(gb && g && phi && c && d_b && n_gauss > 0 && (!gdb || (dd && dd_b)), "smear_bwd: bad arguments");
    hipLaunchKernelGGL(smear_bwd_kernel, dim3(nblk(n_edges * 16)), dim3(256), 0, (hipStream_t)stream, gdb, gb, g, phi,
                       dd, c, (long long)n_edges, n_gauss, d_b, dd_b);
    MDG_CHECK_LAUNCH("smear_bwd_kernel");
    return MDG_OK;
}
