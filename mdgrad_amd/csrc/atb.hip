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

// One wave = one 16-row strip of C (16 values of m) x up to NT column tiles, for one K-slab: the A
// fragment is loaded once per 4 rows and reused by every column tile; 8 rows are in flight per
// iteration (two independent accumulator sets hide the 40-cycle dependent MFMA latency).
// (A2, B2): optional second pair of the same shapes, C = A^T B + A2^T B2 -- the primal + tangent halves of a weight
// gradient in one launch instead of two products and an add.
template <int NT>
__global__ __launch_bounds__(256) void atb_partial_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                          const float* __restrict__ A2, const float* __restrict__ B2,
                                                          long long E, int M, int N, long long slab,
                                                          float* __restrict__ partial) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int tilesM = (M + 15) / 16;
    const int strips_n = ((N + 15) / 16 + NT - 1) / NT;            // groups of NT column tiles
    const int unit = blockIdx.x * 4 + wid;
    if (unit >= tilesM * strips_n) return;
    const int tm = unit / strips_n, tn0 = (unit % strips_n) * NT;
    const int split = blockIdx.y;
    const long long k0 = (long long)split * slab, k1 = min(E, k0 + slab);
    const int li = lane & 15, lk = lane >> 4;
    const int am = tm * 16 + li;
    const bool aok = am < M;
    f32x4 acc0[NT], acc1[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc0[t] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int pass = 0; pass < (A2 ? 2 : 1); ++pass) {
        const float* __restrict__ Ap = pass ? A2 : A;
        const float* __restrict__ Bp = pass ? B2 : B;
        for (long long k = k0; k < k1; k += 8) {
            const long long r0 = k + lk, r1 = k + 4 + lk;
            const bool ok0 = r0 < k1, ok1 = r1 < k1;
            const float a0 = (aok && ok0) ? Ap[r0 * M + am] : 0.f;
            const float a1 = (aok && ok1) ? Ap[r1 * M + am] : 0.f;
            float b0[NT], b1[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int bn = (tn0 + t) * 16 + li;
                b0[t] = (bn < N && ok0) ? Bp[r0 * N + bn] : 0.f;
                b1[t] = (bn < N && ok1) ? Bp[r1 * N + bn] : 0.f;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0[t], acc0[t], 0, 0, 0);
                acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1[t], acc1[t], 0, 0, 0);
            }
        }
    }
    // C layout: col = lane & 15, row = (lane >> 4) * 4 + r
    float* out = partial + (size_t)split * M * N;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = (tn0 + t) * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = tm * 16 + lk * 4 + r;
            if (row < M && col < N) out[(size_t)row * N + col] = acc0[t][r] + acc1[t][r];
        }
    }
}

// C[t] = sum_p partial[p][t]: 4 lanes per output element walk the splits, fixed combine order
__global__ void atb_reduce_kernel(const float* __restrict__ partial, int splits, int MN, float* __restrict__ C) {
    const int t = (blockIdx.x * blockDim.x + threadIdx.x) >> 2, sub = threadIdx.x & 3;
    float s = 0.f;
    if (t < MN)
        for (int p = sub; p < splits; p += 4) s += partial[(size_t)p * MN + t];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (t < MN && sub == 0) C[t] = s;
}

constexpr int ATB_NT = 2;

int atb_units(int M, int N) { return ((M + 15) / 16) * (((N + 15) / 16 + ATB_NT - 1) / ATB_NT); }

int atb_splits(long long E, int M, int N) {
    const int tiles = atb_units(M, N);
    long long want = (1024 + tiles - 1) / tiles;          // ~1024 waves in flight
    const long long maxs = (E + 511) / 512;               // at least 512 rows per slab
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    if (want > 256) want = 256;
    return (int)want;
}

}  // namespace

extern "C" int64_t mdg_atb_workspace(int64_t n_rows, int m, int n) {
    if (n_rows <= 0 || m <= 0 || n <= 0) return 0;
    return (int64_t)atb_splits(n_rows, m, n) * m * n;
}

extern "C" int mdg_atb2(const float* A, const float* B, const float* A2, const float* B2, int64_t n_rows, int m, int n,
                        float* C, float* workspace, void* stream);

extern "C" int mdg_atb(const float* A, const float* B, int64_t n_rows, int m, int n, float* C, float* workspace,
                       void* stream) {
    return mdg_atb2(A, B, nullptr, nullptr, n_rows, m, n, C, workspace, stream);
}

extern "C" int mdg_atb2(const float* A, const float* B, const float* A2, const float* B2, int64_t n_rows, int m, int n,
                        float* C, float* workspace, void* stream) {
    MDG_CHECK_ARG(m > 0 && n > 0 && n_rows >= 0, "atb: bad sizes");
    MDG_CHECK_ARG(C && (n_rows == 0 || (A && B && workspace)), "atb: null buffer");
    MDG_CHECK_ARG((A2 == nullptr) == (B2 == nullptr), "atb: the second pair needs both operands");
    hipStream_t st = (hipStream_t)stream;
    if (n_rows == 0) {
        if (hipMemsetAsync(C, 0, sizeof(float) * (size_t)m * n, st) != hipSuccess) { mdg_set_error("atb: memset failed"); return MDG_ELAUNCH; }
        return MDG_OK;
    }
    const int splits = atb_splits(n_rows, m, n);
    long long slab = (n_rows + splits - 1) / splits;
    slab = (slab + 7) / 8 * 8;
    const int tiles = atb_units(m, n);
    dim3 grid((tiles + 3) / 4, (unsigned)((n_rows + slab - 1) / slab));
    hipLaunchKernelGGL(atb_partial_kernel<ATB_NT>, grid, dim3(256), 0, st, A, B, A2, B2, (long long)n_rows, m, n, slab, workspace);
    hipLaunchKernelGGL(atb_reduce_kernel, dim3((m * n * 4 + 255) / 256), dim3(256), 0, st, workspace, (int)grid.y, m * n, C);
    MDG_CHECK_LAUNCH("atb kernels");
    return MDG_OK;
}
