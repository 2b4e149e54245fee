// Tall-skinny  C[M,N] = A[E,M]^T B[E,N]  (E ~ 1e5..1e6 edges, M,N <= a few hundred): the weight-
// gradient contractions of the edge-wise Dense layers (autograd of nff/nn/layers.py:86-134 applied
// to [E, .] tensors) that appear in the adjoint's parameter vjp.  Library GEMMs have no split-K
// for this shape and run it on M*N/1024 workgroups (measured 0.43 ms per call at E = 2.3e5, 48 % of
// the SchNet step); here every wave owns one 16x16 output tile of one K-slab on the f32 MFMA
// (v_mfma_f32_16x16x4_f32, exact f32), partial tiles go to a [splits, M, N] buffer and a second
// kernel adds the splits in order (deterministic, no atomics).
#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void atb_partial_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                          long long E, int M, int N, long long slab,
                                                          float* __restrict__ partial) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int tilesN = (N + 15) / 16, tilesM = (M + 15) / 16;
    const int tile = blockIdx.x * 4 + wid;
    if (tile >= tilesM * tilesN) return;
    const int tm = tile / tilesN, tn = tile % tilesN;
    const int split = blockIdx.y;
    const long long k0 = (long long)split * slab, k1 = min(E, k0 + slab);
    const int li = lane & 15, lk = lane >> 4;
    const int am = tm * 16 + li, bn = tn * 16 + li;
    const bool aok = am < M, bok = bn < N;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    long long k = k0;
    // two independent accumulators (the 16x16x4 MFMA has a 40-cycle dependent latency, 32-cycle issue)
    for (; k + 8 <= k1; k += 8) {
        const long long r0 = k + lk, r1 = k + 4 + lk;
        const float a0 = aok ? A[r0 * M + am] : 0.f, b0 = bok ? B[r0 * N + bn] : 0.f;
        const float a1 = aok ? A[r1 * M + am] : 0.f, b1 = bok ? B[r1 * N + bn] : 0.f;
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
    }
    for (; k < k1; k += 4) {
        const long long r = k + lk;
        const bool rok = r < k1;
        const float a = (aok && rok) ? A[r * M + am] : 0.f, b = (bok && rok) ? B[r * N + bn] : 0.f;
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
    }
    // C layout: col = lane & 15, row = (lane >> 4) * 4 + r
    float* out = partial + (size_t)split * M * N;
    const int col = tn * 16 + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = tm * 16 + lk * 4 + r;
        if (row < M && col < N) out[(size_t)row * N + col] = acc0[r] + acc1[r];
    }
}

__global__ void atb_reduce_kernel(const float* __restrict__ partial, int splits, int MN, float* __restrict__ C) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= MN) return;
    float s = 0.f;
    for (int p = 0; p < splits; ++p) s += partial[(size_t)p * MN + t];
    C[t] = s;
}

int atb_splits(long long E, int M, int N) {
    const int tiles = ((M + 15) / 16) * ((N + 15) / 16);
    long long want = (2048 + tiles - 1) / tiles;          // ~2048 waves in flight
    const long long maxs = (E + 255) / 256;               // at least 256 rows per slab
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    if (want > 1024) want = 1024;
    return (int)want;
}

}  // namespace

extern "C" int64_t mdg_atb_workspace(int64_t n_rows, int m, int n) {
    if (n_rows <= 0 || m <= 0 || n <= 0) return 0;
    return (int64_t)atb_splits(n_rows, m, n) * m * n;
}

extern "C" int mdg_atb(const float* A, const float* B, int64_t n_rows, int m, int n, float* C, float* workspace,
                       void* stream) {
    MDG_CHECK_ARG(m > 0 && n > 0 && n_rows >= 0, "atb: bad sizes");
    MDG_CHECK_ARG(C && (n_rows == 0 || (A && B && workspace)), "atb: null buffer");
    hipStream_t st = (hipStream_t)stream;
    if (n_rows == 0) {
        if (hipMemsetAsync(C, 0, sizeof(float) * (size_t)m * n, st) != hipSuccess) { mdg_set_error("atb: memset failed"); return MDG_ELAUNCH; }
        return MDG_OK;
    }
    const int splits = atb_splits(n_rows, m, n);
    long long slab = (n_rows + splits - 1) / splits;
    slab = (slab + 7) / 8 * 8;
    const int tiles = ((m + 15) / 16) * ((n + 15) / 16);
    dim3 grid((tiles + 3) / 4, (unsigned)((n_rows + slab - 1) / slab));
    hipLaunchKernelGGL(atb_partial_kernel, grid, dim3(256), 0, st, A, B, (long long)n_rows, m, n, slab, workspace);
    hipLaunchKernelGGL(atb_reduce_kernel, dim3((m * n + 255) / 256), dim3(256), 0, st, workspace, (int)grid.y, m * n, C);
    MDG_CHECK_LAUNCH("atb kernels");
    return MDG_OK;
}
