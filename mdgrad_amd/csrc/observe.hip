// Velocity observables as fused reductions over the saved trajectory v_t[T, N, 3]:
//   vacf         <v(s + t) . v(s)> as the MEAN over frames, atoms and components for lags t = 0 .. L-1
//                (torchmd/observable.py:153-163: one `(vel[t:] * vel[:-t]).mean()` per lag = L elementwise products
//                + L full reductions in the reference) and its gradient w.r.t. v_t;
//   temperature  2 KE / N_dof per frame, N_dof = N dim (torchmd/thermo.py:57-66 on every frame at once).
// HBM-bound single passes; fixed-order partial sums (no atomics): bitwise reproducible.
#include "common.hpp"

namespace {

constexpr int OB_BLOCKS = 64;     // partial sums per lag

// partial[t][b] = sum over this block's slice of idx in [0, (T - t) M) of v[idx + t M] v[idx]
__global__ __launch_bounds__(256) void vacf_partial_kernel(const float* __restrict__ v, int T, long long M,
                                                           float* __restrict__ partial) {
    __shared__ float red[32];
    const int t = blockIdx.y;
    const long long n = (long long)(T - t) * M, shift = (long long)t * M;
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        s = fmaf(v[i + shift], v[i], s);
    s = block_sum(s, red);
    if (threadIdx.x == 0) partial[(size_t)t * gridDim.x + blockIdx.x] = s;
}

__global__ void vacf_finish_kernel(const float* __restrict__ partial, int nb, int T, long long M, int L,
                                   float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L) return;
    double s = 0.0;
    for (int b = 0; b < nb; ++b) s += (double)partial[(size_t)t * nb + b];
    out[t] = (float)(s / ((double)(T - t) * (double)M));
}

// gv[s][e] = sum_t w_t ( [s + t < T] v[s + t][e] + [s - t >= 0] v[s - t][e] ),  w_t = g_t / ((T - t) M)
__global__ __launch_bounds__(256) void vacf_bwd_kernel(const float* __restrict__ v, const float* __restrict__ g, int T,
                                                       long long M, int L, float* __restrict__ gv) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)T * M) return;
    const int s = (int)(i / M);
    float acc = 0.f;
    for (int t = 0; t < L; ++t) {
        const float w = g[t] / ((float)(T - t) * (float)M);
        if (s + t < T) acc = fmaf(w, v[i + (long long)t * M], acc);
        if (s - t >= 0) acc = fmaf(w, v[i - (long long)t * M], acc);
    }
    gv[i] = acc;
}

// one workgroup per frame: out[f] = sum_n m_n |v_n|^2 / dof
__global__ __launch_bounds__(256) void temperature_kernel(const float* __restrict__ v, const float* __restrict__ mass,
                                                          int N, float inv_dof, float* __restrict__ out) {
    __shared__ float red[32];
    const float* vf = v + (size_t)blockIdx.x * N * 3;
    float s = 0.f;
    for (int e = threadIdx.x; e < 3 * N; e += blockDim.x) { const float x = vf[e]; s = fmaf(mass[e / 3] * x, x, s); }
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s * inv_dof;
}

}  // namespace

extern "C" int64_t mdg_vacf_workspace(int n_lags) { return n_lags > 0 ? (int64_t)n_lags * OB_BLOCKS : 0; }

extern "C" int mdg_vacf_fwd(const float* v, int n_frames, int64_t n_dof_per_frame, int n_lags, float* out,
                            float* workspace, void* stream) {
    MDG_CHECK_ARG(v && out && workspace && n_frames > 0 && n_dof_per_frame > 0, "vacf_fwd: bad arguments");
    MDG_CHECK_ARG(n_lags >= 1 && n_lags <= n_frames, "vacf_fwd: 1 <= lags <= frames (got %d, %d)", n_lags, n_frames);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(vacf_partial_kernel, dim3(OB_BLOCKS, n_lags), dim3(256), 0, st, v, n_frames,
                       (long long)n_dof_per_frame, workspace);
    hipLaunchKernelGGL(vacf_finish_kernel, dim3((n_lags + 63) / 64), dim3(64), 0, st, workspace, OB_BLOCKS, n_frames,
                       (long long)n_dof_per_frame, n_lags, out);
    MDG_CHECK_LAUNCH("vacf kernels");
    return MDG_OK;
}

extern "C" int mdg_vacf_bwd(const float* v, const float* g_out, int n_frames, int64_t n_dof_per_frame, int n_lags,
                            float* g_v, void* stream) {
    MDG_CHECK_ARG(v && g_out && g_v && n_frames > 0 && n_dof_per_frame > 0 && n_lags >= 1 && n_lags <= n_frames,
                  "vacf_bwd: bad arguments");
    const long long tot = (long long)n_frames * n_dof_per_frame;
    hipLaunchKernelGGL(vacf_bwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, v, g_out,
                       n_frames, (long long)n_dof_per_frame, n_lags, g_v);
    MDG_CHECK_LAUNCH("vacf_bwd_kernel");
    return MDG_OK;
}

extern "C" int mdg_temperature(const float* v, const float* mass, int n_frames, int n_atoms, float n_dof, float* out,
                               void* stream) {
    MDG_CHECK_ARG(v && mass && out && n_frames > 0 && n_atoms > 0 && n_dof > 0.f, "temperature: bad arguments");
    hipLaunchKernelGGL(temperature_kernel, dim3(n_frames), dim3(256), 0, (hipStream_t)stream, v, mass, n_atoms, 1.0f / n_dof,
                       out);
    MDG_CHECK_LAUNCH("temperature_kernel");
    return MDG_OK;
}
