// K11 / K12: node-level Dense layers of the SchNet block with their epilogues fused
// (nff/nn/layers.py:86-134 `Dense` = act(x W^T + b), nff/nn/activations.py:5-11 shifted softplus; the
//  update MLP and the residual of nff/nn/modules.py:543-547 / nff/nn/models/schnet.py:149-151, the readout's first
//  Linear, and the transposed products x W of the reverse sweeps in mdgrad_amd/nn/analytic.py):
//
//     z   = x B (+ bias)            B[k][m] = W[m][k] (Linear layout, forward)  or  W[k][m] (reverse sweeps)
//     out = act(z) (* mul) (+ res)  act = identity | shifted softplus (then sig = sigmoid(z) is stored as well)
//
// One launch serves up to two inputs that share the weight (DUAL): the primal rows and their forward-mode tangent
// (z1 = x1 B ; out1 = act'(z0) z1 (+ res1)), or the two adjoints of the dual reverse sweep -- one set of B
// fragments from LDS feeds both accumulator sets.  v_mfma_f32_16x16x4_f32 (exact f32): persistent workgroups
// stage one chunk of <= 128 output columns of the weight (k <= 256 per launch; wider layers run as k-slabs that
// hand their partial sums on through the output buffer) in LDS once and walk 64-row tiles (16 rows
// per wave); LDS rows are permuted so that the A operand is gathered as 16-byte vectors (k-step 4 q + c of a
// 64-chunk <-> k = 16 q + 4 lk + c).
#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr float LOG2E_D = 1.4426950408889634f;
constexpr float LN2_D = 0.69314718055994531f;
constexpr int DN_KC = 64;          // k chunk
constexpr int DN_MC = 128;         // output columns per workgroup
constexpr int DN_SB = 144;         // LDS row stride of the weight chunk (16 mod 32)

struct DenseProb {
    const float* x;      // [N, K]
    const float* bias;   // [M] or null        (row 0 only)
    const float* mul;    // [N, M] or null     (row 0 only)
    const float* res;    // [N, M] or null
    float* out;          // [N, M]
    float* sig;          // [N, M] or null     (row 0, act == 1)
    const float* pre;    // [N, M] or null: added to z before bias / activation (the partial sums of earlier k chunks)
};

struct DenseArgs {
    DenseProb p[2];      // p[1] = the tangent / second adjoint sharing the weight (DUAL)
    const float* W;
    int trans;           // 0: B[k][m] = W[m*K + k] ; 1: B[k][m] = W[k*M + m]
    int act;             // 0 identity, 1 shifted softplus: out0 = ssp(z0), sig0 = sigmoid(z0), out1 = sigmoid(z0) z1
    int N, K, M;         // K: the k range of THIS launch (<= 256)
    int ldx, ldw;        // floats per row of x and (trans = 0) of W: the layer's full k
};

__device__ __forceinline__ void load_a(const float* __restrict__ x, int arow, bool aok, int K, int ldx, bool vec, int kc, int lk,
                                       float (&af)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k0 = kc + 16 * q + 4 * lk;
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (aok) {
            if (vec && k0 + 4 <= K) v = *reinterpret_cast<const float4*>(x + (size_t)arow * ldx + k0);
            else {
                const float* px = x + (size_t)arow * ldx;
                if (k0 < K) v.x = px[k0];
                if (k0 + 1 < K) v.y = px[k0 + 1];
                if (k0 + 2 < K) v.z = px[k0 + 2];
                if (k0 + 3 < K) v.w = px[k0 + 3];
            }
        }
        af[4 * q] = v.x; af[4 * q + 1] = v.y; af[4 * q + 2] = v.z; af[4 * q + 3] = v.w;
    }
}

// VEC: the chunk has 4 or 8 column tiles (64 / 128 columns) -- LDS column (nt, li) then holds output column
// li * NT + nt, so a lane owns NT consecutive columns of its 4 rows and the epilogue moves 16-byte vectors.
template <bool DUAL, int NT>          // NT = 0: general (scalar epilogue, natural column order)
__global__ __launch_bounds__(256) void dense_kernel(const DenseArgs A) {
    extern __shared__ __attribute__((aligned(16))) float bs[];          // [Kpad][DN_SB]: the whole weight chunk, staged once
    const DenseProb P0 = A.p[0], P1 = A.p[1];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int m_lo = blockIdx.y * DN_MC;
    const int Mc = min(DN_MC, A.M - m_lo);
    const int ntiles = NT ? NT : (Mc + 15) >> 4;
    const int Kpad = (A.K + DN_KC - 1) / DN_KC * DN_KC;
    const int Mp = ntiles * 16;
    // physical row kc + (4 q + c) * 4 + lk holds k = kc + 16 q + 4 lk + c  (kc = 64-chunk base)
    auto put = [&](int k, int m, float v) {
        const int kk = k & (DN_KC - 1);
        const int prow = (k - kk) + (4 * (kk >> 4) + (kk & 3)) * 4 + ((kk & 15) >> 2);
        const int col = NT ? (m % NT) * 16 + m / NT : m;
        bs[prow * DN_SB + col] = v;
    };
    if (!A.trans && (A.K & 3) == 0 && (A.ldw & 3) == 0) {
        // Linear layout W[m][k]: 16-byte loads along k, several in flight
        const int kq = Kpad >> 2;
#pragma unroll 4
        for (int t = tid; t < kq * Mp; t += 256) {
            const int m = t / kq, k = (t % kq) * 4;
            float4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < A.K && m < Mc) v = *reinterpret_cast<const float4*>(A.W + (size_t)(m_lo + m) * A.ldw + k);
            put(k, m, v.x); put(k + 1, m, v.y); put(k + 2, m, v.z); put(k + 3, m, v.w);
        }
    } else {
#pragma unroll 4
        for (int t = tid; t < Kpad * Mp; t += 256) {
            int k, m;
            if (A.trans) { k = t / Mp; m = t % Mp; } else { m = t / Kpad; k = t % Kpad; }
            float v = 0.f;
            if (k < A.K && m < Mc) v = A.trans ? A.W[(size_t)k * A.M + m_lo + m] : A.W[(size_t)(m_lo + m) * A.ldw + k];
            put(k, m, v);
        }
    }
    __syncthreads();
    const bool vec = (A.K & 3) == 0 && (A.ldx & 3) == 0;
    const int row_tiles = (A.N + 63) >> 6;
    constexpr int TT = NT ? NT : DN_MC / 16;
    // software pipeline over the (row tile, k chunk) sequence of this workgroup: the A fragments of the NEXT chunk -- the
    // first chunk of the next row tile after the last one -- are requested before the MFMAs of the current chunk, so the
    // global loads overlap the matrix work and the epilogue instead of heading every chunk with a round trip
    float a0[16], a1[16];
    {
        const int arow = (int)blockIdx.x * 64 + wid * 16 + li;
        const bool aok = (int)blockIdx.x < row_tiles && arow < A.N;
        load_a(P0.x, arow, aok, A.K, A.ldx, vec, 0, lk, a0);
        if (DUAL) load_a(P1.x, arow, aok, A.K, A.ldx, vec, 0, lk, a1);
    }
    for (int tile = blockIdx.x; tile < row_tiles; tile += gridDim.x) {
        const int row0 = tile * 64 + wid * 16;
        f32x4 acc0[TT], acc1[DUAL ? TT : 1];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            acc0[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (DUAL) acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int kc = 0; kc < A.K; kc += DN_KC) {
            float n0[16], n1[16];
            {
                const bool more_k = kc + DN_KC < A.K;
                const int ntile = more_k ? tile : tile + (int)gridDim.x;
                const int nrow = ntile * 64 + wid * 16 + li;
                const bool nok = ntile < row_tiles && nrow < A.N;
                load_a(P0.x, nrow, nok, A.K, A.ldx, vec, more_k ? kc + DN_KC : 0, lk, n0);
                if (DUAL) load_a(P1.x, nrow, nok, A.K, A.ldx, vec, more_k ? kc + DN_KC : 0, lk, n1);
            }
            const float* bk = bs + (size_t)kc * DN_SB;
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                if (NT || t < ntiles) {
#pragma unroll
                    for (int ks = 0; ks < 16; ++ks) {
                        const float bfr = bk[(ks * 4 + lk) * DN_SB + t * 16 + li];
                        acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[ks], bfr, acc0[t], 0, 0, 0);
                        if (DUAL) acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[ks], bfr, acc1[t], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) { a0[q] = n0[q]; if (DUAL) a1[q] = n1[q]; }
        }
        if constexpr (NT != 0) {
            // lane: rows row0 + 4 lk + r, columns m_lo + li * NT + [0, NT)
            const int mb = m_lo + li * NT;
            float bv[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) bv[t] = P0.bias ? P0.bias[mb + t] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 4 * lk + r;
                if (row >= A.N) continue;
                const size_t o = (size_t)row * A.M + mb;
                float z[NT], z1[NT], sg[NT];
                float pz[NT], pz1[NT];
#pragma unroll
                for (int v = 0; v < NT / 4; ++v) {
                    float4 q = {0.f, 0.f, 0.f, 0.f}, q1 = q;
                    if (P0.pre) q = *reinterpret_cast<const float4*>(P0.pre + o + 4 * v);
                    if (DUAL && P1.pre) q1 = *reinterpret_cast<const float4*>(P1.pre + o + 4 * v);
                    pz[4 * v] = q.x; pz[4 * v + 1] = q.y; pz[4 * v + 2] = q.z; pz[4 * v + 3] = q.w;
                    pz1[4 * v] = q1.x; pz1[4 * v + 1] = q1.y; pz1[4 * v + 2] = q1.z; pz1[4 * v + 3] = q1.w;
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    z[t] = acc0[t][r] + pz[t] + bv[t];
                    z1[t] = DUAL ? acc1[t][r] + pz1[t] : 0.f;
                    if (A.act == 1) {
                        const float ex = __builtin_amdgcn_exp2f(z[t] * LOG2E_D);
                        const bool big = z[t] > 20.f;
                        const float sp = __builtin_amdgcn_logf(1.0f + ex) * LN2_D;
                        sg[t] = big ? 1.0f : ex * __builtin_amdgcn_rcpf(1.0f + ex);
                        z[t] = (big ? z[t] : sp) - LN2_D;
                        z1[t] *= sg[t];
                    }
                }
#pragma unroll
                for (int v = 0; v < NT / 4; ++v) {
                    if (A.act == 1 && P0.sig)
                        *reinterpret_cast<float4*>(P0.sig + o + 4 * v) = make_float4(sg[4 * v], sg[4 * v + 1], sg[4 * v + 2], sg[4 * v + 3]);
                    if (P0.mul) {
                        const float4 q = *reinterpret_cast<const float4*>(P0.mul + o + 4 * v);
                        z[4 * v] *= q.x; z[4 * v + 1] *= q.y; z[4 * v + 2] *= q.z; z[4 * v + 3] *= q.w;
                    }
                    if (P0.res) {
                        const float4 q = *reinterpret_cast<const float4*>(P0.res + o + 4 * v);
                        z[4 * v] += q.x; z[4 * v + 1] += q.y; z[4 * v + 2] += q.z; z[4 * v + 3] += q.w;
                    }
                    *reinterpret_cast<float4*>(P0.out + o + 4 * v) = make_float4(z[4 * v], z[4 * v + 1], z[4 * v + 2], z[4 * v + 3]);
                    if (DUAL) {
                        if (P1.res) {
                            const float4 q = *reinterpret_cast<const float4*>(P1.res + o + 4 * v);
                            z1[4 * v] += q.x; z1[4 * v + 1] += q.y; z1[4 * v + 2] += q.z; z1[4 * v + 3] += q.w;
                        }
                        *reinterpret_cast<float4*>(P1.out + o + 4 * v) =
                            make_float4(z1[4 * v], z1[4 * v + 1], z1[4 * v + 2], z1[4 * v + 3]);
                    }
                }
            }
        } else {
            // general: accumulator layout, row = row0 + 4 lk + r, column = m_lo + 16 t + li
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                if (t >= ntiles) continue;
                const int m = t * 16 + li;
                if (m >= Mc) continue;
                const float b0 = P0.bias ? P0.bias[m_lo + m] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = row0 + 4 * lk + r;
                    if (row >= A.N) continue;
                    const size_t o = (size_t)row * A.M + m_lo + m;
                    float z = acc0[t][r] + (P0.pre ? P0.pre[o] : 0.f) + b0;
                    float z1 = DUAL ? acc1[t][r] + (P1.pre ? P1.pre[o] : 0.f) : 0.f;
                    if (A.act == 1) {
                        const float ex = __builtin_amdgcn_exp2f(z * LOG2E_D);
                        const bool big = z > 20.f;
                        const float sp = __builtin_amdgcn_logf(1.0f + ex) * LN2_D;
                        const float sg = big ? 1.0f : ex * __builtin_amdgcn_rcpf(1.0f + ex);
                        if (P0.sig) P0.sig[o] = sg;
                        z = (big ? z : sp) - LN2_D;
                        z1 *= sg;                                   // tangent of the activation
                    }
                    if (P0.mul) z *= P0.mul[o];
                    if (P0.res) z += P0.res[o];
                    P0.out[o] = z;
                    if (DUAL) {
                        if (P1.res) z1 += P1.res[o];
                        P1.out[o] = z1;
                    }
                }
            }
        }
    }
}

}  // namespace

extern "C" int mdg_dense(const float* W, int trans, int act, int n_rows, int k, int m,
                         const float* x0, const float* bias0, const float* mul0, const float* res0, float* out0,
                         float* sig0, const float* x1, const float* res1, float* out1, void* stream) {
    MDG_CHECK_ARG(n_rows >= 0 && k > 0 && m > 0, "dense: bad sizes");
    if (n_rows == 0) return MDG_OK;
    MDG_CHECK_ARG(W && x0 && out0 && (!x1 || out1), "dense: null buffer");
    MDG_CHECK_ARG(act == 0 || act == 1, "dense: act must be 0 (identity) or 1 (shifted softplus)");
    MDG_CHECK_ARG((((uintptr_t)x0 | (uintptr_t)x1) & 15) == 0, "dense: inputs must be 16-byte aligned");
    const int row_tiles = (n_rows + 63) / 64;
    // persistent workgroups: as many per CU as the LDS-resident weight chunk allows (<= 4), so that several row tiles are in
    // flight per CU -- with one workgroup per CU the kernel ran at ~2 TB/s on [32768 x 128] rows (latency-bound loads)
    const size_t lds_max = sizeof(float) * (size_t)(((k < 256 ? k : 256) + DN_KC - 1) / DN_KC * DN_KC) * DN_SB;
    int per_cu = (int)((size_t)(160 * 1024) / (lds_max + 1024));
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    const int gx_max = 256 * per_cu;
    dim3 grid(row_tiles < gx_max ? row_tiles : gx_max, (m + DN_MC - 1) / DN_MC);
    // 16-byte epilogue when every column chunk is 64 or 128 wide and all row pointers are 16-byte aligned
    const uintptr_t al = (uintptr_t)mul0 | (uintptr_t)res0 | (uintptr_t)out0 | (uintptr_t)sig0 | (uintptr_t)res1 |
                         (uintptr_t)out1 | (uintptr_t)bias0;
    const int nt = ((al & 15) == 0 && m % 64 == 0 && (m <= DN_MC ? true : m % DN_MC == 0)) ? (m >= DN_MC ? 8 : 4) : 0;
    hipStream_t st = (hipStream_t)stream;
    // The weight chunk of a launch (<= 128 output columns x all its k) lives in LDS, which holds k <= 256: wider layers
    // (n_atom_basis / n_filters 512, demo/fit_rdf_gnn.py:16-19) run as k-slabs of 256 -- the earlier slabs leave their partial
    // sums z in out0 / out1, the last one adds them before bias and activation (`pre`).
    constexpr int DN_KMAX = 256;
    for (int k0 = 0; k0 < k; k0 += DN_KMAX) {
        const int kc = k - k0 < DN_KMAX ? k - k0 : DN_KMAX;
        const bool first = k0 == 0, last = k0 + kc >= k;
        DenseArgs a{};
        a.p[0] = DenseProb{x0 + k0, last ? bias0 : nullptr, last ? mul0 : nullptr, last ? res0 : nullptr, out0, last ? sig0 : nullptr,
                           first ? nullptr : out0};
        a.p[1] = DenseProb{x1 ? x1 + k0 : nullptr, nullptr, nullptr, last ? res1 : nullptr, out1, nullptr, first ? nullptr : out1};
        a.W = trans ? W + (size_t)k0 * m : W + k0;
        a.trans = trans; a.act = last ? act : 0; a.N = n_rows; a.K = kc; a.M = m; a.ldx = k; a.ldw = k;
        const size_t lds = sizeof(float) * (size_t)((kc + DN_KC - 1) / DN_KC * DN_KC) * DN_SB;
#define MDG_DENSE(D_, N_) hipLaunchKernelGGL((dense_kernel<D_, N_>), grid, dim3(256), lds, st, a)
        if (x1) { if (nt == 8) MDG_DENSE(true, 8); else if (nt == 4) MDG_DENSE(true, 4); else MDG_DENSE(true, 0); }
        else { if (nt == 8) MDG_DENSE(false, 8); else if (nt == 4) MDG_DENSE(false, 4); else MDG_DENSE(false, 0); }
    }
#undef MDG_DENSE
    MDG_CHECK_LAUNCH("dense_kernel");
    return MDG_OK;
}
