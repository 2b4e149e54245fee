// K9: continuous-filter generator  W[e,:] = Dense2( ssp( Dense1( smear(d_e) ) ) )
// (nff/nn/modules.py:531-541 with GaussianSmearing nff/nn/layers.py:14-31, Dense :86-134,
//  shifted_softplus nff/nn/activations.py:5-11), fused into one kernel on the matrix cores.
//
// This is the one GEMM-shaped piece of the hot path (K = G = 12..64 radial basis functions,
// N = G then F filters, M = E edges), so it goes on MFMA: v_mfma_f32_16x16x4_f32 -- f32 in,
// f32 accumulate, bitwise a k-ordered fmaf chain, which keeps fp32 parity with the reference's
// F.linear.  Per workgroup: 64 edges (one 16-row tile per wave) x one chunk of <= 128 filters.
//   layer 1: the A fragment (smearing values) is computed in registers -- the [E,G] Gaussian
//            matrix never exists; B = W1^T from LDS; bias + shifted softplus in the epilogue;
//   layer 2: H1 goes through LDS once to turn the C layout into the A layout; B = W2^T chunk
//            from LDS; bias in the epilogue; the only HBM traffic is d[E] in and W[E,F] out.
// LDS strides are chosen so the 16x4 operand fetches are bank-conflict free in each half-wave
// (B: stride = 16 mod 32 floats; A: stride = 2 mod 4 floats).
#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FT_TM = 64;        // edges per workgroup
constexpr int FT_FCH = 128;      // filters per workgroup (grid.y chunks)
constexpr int FT_GMAX = 64;

__host__ __device__ inline int stride16mod32(int n) {   // smallest s >= n with s % 32 == 16
    int s = (n + 31) / 32 * 32 - 16;
    if (s < n) s += 32;
    return s;
}

__device__ __forceinline__ float ssp(float x) {          // softplus(x) - ln 2, torch threshold 20
    // log1p(e^x) as log2(1 + 2^(x log2 e)) * ln 2 on the hardware exp2/log2 (abs. error ~1e-7)
    const float ex = __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
    const float sp = x > 20.f ? x : __builtin_amdgcn_logf(1.0f + ex) * 0.69314718055994531f;
    return sp - 0.69314718055994531f;
}

template <int GP>
__global__ __launch_bounds__(256) void cfconv_filter_kernel(
    const float* __restrict__ d, long long E, const float* __restrict__ mu, const float* __restrict__ width,
    int G, const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
    const float* __restrict__ b2, int F, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int Gp = GP;                 // G padded to a multiple of 16 (compile time: no run-time guards in the MFMA loops)
    const int S1 = stride16mod32(Gp);
    const int f_lo = blockIdx.y * FT_FCH;
    const int Fc = min(FT_FCH, F - f_lo);
    const int Fcp = (Fc + 15) / 16 * 16;
    const int S2 = stride16mod32(Fcp);
    const int SA = Gp + 2;
    const int SO = Fcp + 4;                // output staging stride (rows stay 16-B aligned)
    float* w1s = sm;                       // [Gp][S1]   B1[k][j] = W1[j][k]
    float* w2s = w1s + Gp * S1;            // [Gp][S2]   B2[k][j] = W2[f_lo + j][k]
    float* h1s = w2s + Gp * S2;            // [4 waves][16][SA]
    float* mus = h1s + 4 * 16 * SA;        // [Gp] centres, [Gp] coeff, [Gp] b1, [Fcp] b2
    float* cfs = mus + Gp;
    float* b1s = cfs + Gp;
    float* b2s = b1s + Gp;
    float* outs = b2s + Fcp;               // [4 waves][16][SO] output staging for coalesced row stores
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

    // weights are staged ONCE per (persistent) workgroup
    for (int t = tid; t < Gp * Gp; t += 256) {
        const int k = t / Gp, j = t % Gp;
        w1s[k * S1 + j] = (k < G && j < G) ? W1[j * G + k] : 0.f;
    }
    for (int t = tid; t < Gp * Fcp; t += 256) {
        const int k = t % Gp, j = t / Gp;              // consecutive threads read consecutive k of one row
        w2s[k * S2 + j] = (k < G && j < Fc) ? W2[(size_t)(f_lo + j) * G + k] : 0.f;
    }
    for (int k = tid; k < Gp; k += 256) {
        mus[k] = k < G ? mu[k] : 0.f;
        const float w = k < G ? width[k] : 1.f;
        cfs[k] = k < G ? -0.5f / (w * w) * 1.4426950408889634f : 0.f;   // exp(c x^2) = exp2(c log2e x^2)
        b1s[k] = k < G ? b1[k] : 0.f;
    }
    for (int j = tid; j < Fcp; j += 256) b2s[j] = j < Fc ? b2[f_lo + j] : 0.f;
    __syncthreads();

    const int li = lane & 15, lk = lane >> 4;
    float* h1w = h1s + wid * 16 * SA;
    float* ow = outs + wid * 16 * SO;
    const long long ntiles = (E + FT_TM - 1) / FT_TM;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long e0 = tile * FT_TM + wid * 16;
        // ---- layer 1: A[i][k] = exp(c_k (d_i - mu_k)^2) computed in registers
        const long long ea = e0 + li;
        const float da = ea < E ? d[ea] : 0.f;
        // (padded k >= G: the matching rows of W1^T in LDS are zero, so the value there is irrelevant)
        float afrag[GP / 4];
#pragma unroll
        for (int ks = 0; ks < GP / 4; ++ks) {
            const int k = ks * 4 + lk;
            const float x = da - mus[k];
            afrag[ks] = __builtin_amdgcn_exp2f(cfs[k] * x * x);
        }
#pragma unroll
        for (int nt = 0; nt < GP / 16; ++nt) {
            float bfrag[GP / 4];
#pragma unroll
            for (int ks = 0; ks < GP / 4; ++ks) bfrag[ks] = w1s[(ks * 4 + lk) * S1 + nt * 16 + li];
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < GP / 4; ++ks)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[ks], bfrag[ks], acc, 0, 0, 0);
            // C layout: col = lane & 15, row = (lane >> 4) * 4 + r.  Padded columns: zero weights and
            // zero bias give ssp(0) = 0.
            const int col = nt * 16 + li;
            const float bias = b1s[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) h1w[(lk * 4 + r) * SA + col] = ssp(acc[r] + bias);
        }
        // (h1w / ow are private to the wave: program order + the LDS counter suffice, no barrier)
        // ---- layer 2: A[i][k] = H1[i][k] from LDS
#pragma unroll
        for (int ks = 0; ks < GP / 4; ++ks) afrag[ks] = h1w[li * SA + ks * 4 + lk];
        for (int nt = 0; nt < Fcp / 16; ++nt) {
            float bfrag[GP / 4];
#pragma unroll
            for (int ks = 0; ks < GP / 4; ++ks) bfrag[ks] = w2s[(ks * 4 + lk) * S2 + nt * 16 + li];
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < GP / 4; ++ks)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[ks], bfrag[ks], acc, 0, 0, 0);
            const int col = nt * 16 + li;
            const float bias = b2s[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) ow[(lk * 4 + r) * SO + col] = acc[r] + bias;
        }
        // ---- coalesced row stores: 16 rows x Fc floats, consecutive lanes -> consecutive columns
        if ((Fc & 3) == 0 && (F & 3) == 0) {
            const int qpr = Fc / 4;                        // float4 per row
            for (int t = lane; t < 16 * qpr; t += 64) {
                const int row = t / qpr, c4 = t % qpr;
                const long long e = e0 + row;
                if (e < E)
                    *reinterpret_cast<float4*>(&out[(size_t)e * F + f_lo + 4 * c4]) =
                        *reinterpret_cast<const float4*>(&ow[row * SO + 4 * c4]);
            }
        } else {
            for (int t = lane; t < 16 * Fc; t += 64) {
                const int row = t / Fc, c = t % Fc;
                const long long e = e0 + row;
                if (e < E) out[(size_t)e * F + f_lo + c] = ow[row * SO + c];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// bf16-operand variant (BASELINE config #5: "bf16 cfconv MFMA"): v_mfma_f32_16x16x32_bf16, fp32
// accumulate, fp32 biases / activation / output.  K = 32 per instruction, so G <= 32 needs ONE MFMA
// per 16x16 output tile (16x the f32 rate): the kernel becomes purely HBM-write bound.  Operands are
// rounded to bf16 (round-to-nearest-even); expect ~3 significant digits on W.
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned int u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

template <int GP>     // GP in {32, 64}: K padded to a multiple of 32
__global__ __launch_bounds__(256) void cfconv_filter_bf16_kernel(
    const float* __restrict__ d, long long E, const float* __restrict__ mu, const float* __restrict__ width,
    int G, const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
    const float* __restrict__ b2, int F, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int KS = GP + 8;             // bf16 row stride (elements): 16-B aligned rows, conflict-free b128 reads
    const int f_lo = blockIdx.y * FT_FCH;
    const int Fc = min(FT_FCH, F - f_lo);
    const int Fcp = (Fc + 15) / 16 * 16;
    const int SO = Fcp + 4;
    unsigned short* w1s = reinterpret_cast<unsigned short*>(sm);      // [GP rows j][KS]  W1[j][k]
    unsigned short* w2s = w1s + GP * KS;                               // [Fcp rows j][KS] W2[f_lo + j][k]
    unsigned short* h1s = w2s + FT_FCH * KS;                           // [4 waves][16][KS]
    float* mus = reinterpret_cast<float*>(h1s + 4 * 16 * KS);          // [GP], then coeff [GP], b1 [GP], b2 [Fcp]
    float* cfs = mus + GP;
    float* b1s = cfs + GP;
    float* b2s = b1s + GP;
    float* outs = b2s + FT_FCH;                                        // [4 waves][16][SO]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int t = tid; t < GP * GP; t += 256) {
        const int j = t / GP, k = t % GP;
        w1s[j * KS + k] = (j < G && k < G) ? f2bf(W1[j * G + k]) : 0;
    }
    for (int t = tid; t < Fcp * GP; t += 256) {
        const int j = t / GP, k = t % GP;
        w2s[j * KS + k] = (j < Fc && k < G) ? f2bf(W2[(size_t)(f_lo + j) * G + k]) : 0;
    }
    for (int k = tid; k < GP; k += 256) {
        mus[k] = k < G ? mu[k] : 0.f;
        const float w = k < G ? width[k] : 1.f;
        cfs[k] = k < G ? -0.5f / (w * w) * 1.4426950408889634f : 0.f;
        b1s[k] = k < G ? b1[k] : 0.f;
    }
    for (int j = tid; j < Fcp; j += 256) b2s[j] = j < Fc ? b2[f_lo + j] : 0.f;
    __syncthreads();

    const int li = lane & 15, lk = lane >> 4;
    unsigned short* h1w = h1s + wid * 16 * KS;
    float* ow = outs + wid * 16 * SO;
    const long long ntiles = (E + FT_TM - 1) / FT_TM;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long e0 = tile * FT_TM + wid * 16;
        const long long ea = e0 + li;
        const float da = ea < E ? d[ea] : 0.f;
        // ---- layer 1: A[i][k = ks*32 + lk*8 + t] = smear(d_i, k) as bf16, in registers
        bf16x8 afrag[GP / 32];
#pragma unroll
        for (int ks = 0; ks < GP / 32; ++ks) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int k = ks * 32 + lk * 8 + t;
                const float x = da - mus[k];
                afrag[ks][t] = (short)f2bf(__builtin_amdgcn_exp2f(cfs[k] * x * x));
            }
        }
#pragma unroll
        for (int nt = 0; nt < GP / 16; ++nt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < GP / 32; ++ks) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(&w1s[(nt * 16 + li) * KS + ks * 32 + lk * 8]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[ks], b, acc, 0, 0, 0);
            }
            const int col = nt * 16 + li;
            const float bias = b1s[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) h1w[(lk * 4 + r) * KS + col] = f2bf(ssp(acc[r] + bias));
        }
        // ---- layer 2
#pragma unroll
        for (int ks = 0; ks < GP / 32; ++ks)
            afrag[ks] = *reinterpret_cast<const bf16x8*>(&h1w[li * KS + ks * 32 + lk * 8]);
        for (int nt = 0; nt < Fcp / 16; ++nt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < GP / 32; ++ks) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(&w2s[(nt * 16 + li) * KS + ks * 32 + lk * 8]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[ks], b, acc, 0, 0, 0);
            }
            const int col = nt * 16 + li;
            const float bias = b2s[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) ow[(lk * 4 + r) * SO + col] = acc[r] + bias;
        }
        if ((Fc & 3) == 0 && (F & 3) == 0) {
            const int qpr = Fc / 4;
            for (int t = lane; t < 16 * qpr; t += 64) {
                const int row = t / qpr, c4 = t % qpr;
                const long long e = e0 + row;
                if (e < E)
                    *reinterpret_cast<float4*>(&out[(size_t)e * F + f_lo + 4 * c4]) =
                        *reinterpret_cast<const float4*>(&ow[row * SO + 4 * c4]);
            }
        } else {
            for (int t = lane; t < 16 * Fc; t += 64) {
                const int row = t / Fc, c = t % Fc;
                const long long e = e0 + row;
                if (e < E) out[(size_t)e * F + f_lo + c] = ow[row * SO + c];
            }
        }
    }
}

}  // namespace

extern "C" int mdg_cfconv_filter(const float* d, int64_t n_edges, const float* mu, const float* width, int n_gauss,
                                 const float* W1, const float* b1, const float* W2, const float* b2,
                                 int n_filters, float* out, void* stream) {
    MDG_CHECK_ARG(n_edges >= 0 && n_gauss > 0 && n_filters > 0, "cfconv_filter: bad sizes");
    MDG_CHECK_ARG(n_gauss <= FT_GMAX, "cfconv_filter: n_gaussians > %d not supported", FT_GMAX);
    if (n_edges == 0) return MDG_OK;
    MDG_CHECK_ARG(d && mu && width && W1 && b1 && W2 && b2 && out, "cfconv_filter: null buffer");
    const int Gp = (n_gauss + 15) / 16 * 16;
    const int Fc = n_filters < FT_FCH ? n_filters : FT_FCH;
    const int Fcp = (Fc + 15) / 16 * 16;
    const size_t lds = sizeof(float) * ((size_t)Gp * stride16mod32(Gp) + (size_t)Gp * stride16mod32(Fcp) +
                                        4 * 16 * (Gp + 2) + 3 * Gp + Fcp + 4 * 16 * (Fcp + 4));
    const long long ntiles = (n_edges + FT_TM - 1) / FT_TM;
    const int chunks = (n_filters + FT_FCH - 1) / FT_FCH;
    // persistent workgroups: ~4 per CU share the work, each stages the weights once
    const long long want = 1024 / chunks > 1 ? 1024 / chunks : 1;
    dim3 grid((unsigned)(ntiles < want ? ntiles : want), chunks);
#define MDG_FILTER_LAUNCH(GP_)                                                                         \
    hipLaunchKernelGGL(cfconv_filter_kernel<GP_>, grid, dim3(256), lds, (hipStream_t)stream, d,        \
                       (long long)n_edges, mu, width, n_gauss, W1, b1, W2, b2, n_filters, out)
    if (Gp == 16) MDG_FILTER_LAUNCH(16);
    else if (Gp == 32) MDG_FILTER_LAUNCH(32);
    else if (Gp == 48) MDG_FILTER_LAUNCH(48);
    else MDG_FILTER_LAUNCH(64);
    MDG_CHECK_LAUNCH("cfconv_filter_kernel");
    return MDG_OK;
}

extern "C" int mdg_cfconv_filter_bf16(const float* d, int64_t n_edges, const float* mu, const float* width,
                                      int n_gauss, const float* W1, const float* b1, const float* W2,
                                      const float* b2, int n_filters, float* out, void* stream) {
    MDG_CHECK_ARG(n_edges >= 0 && n_gauss > 0 && n_filters > 0, "cfconv_filter_bf16: bad sizes");
    MDG_CHECK_ARG(n_gauss <= FT_GMAX, "cfconv_filter_bf16: n_gaussians > %d not supported", FT_GMAX);
    if (n_edges == 0) return MDG_OK;
    MDG_CHECK_ARG(d && mu && width && W1 && b1 && W2 && b2 && out, "cfconv_filter_bf16: null buffer");
    const int GP = n_gauss <= 32 ? 32 : 64;
    const int KS = GP + 8;
    const int Fc = n_filters < FT_FCH ? n_filters : FT_FCH;
    const int Fcp = (Fc + 15) / 16 * 16;
    const size_t lds = sizeof(unsigned short) * ((size_t)GP * KS + (size_t)FT_FCH * KS + 4 * 16 * KS) +
                       sizeof(float) * (3 * (size_t)GP + FT_FCH + 4 * 16 * (Fcp + 4));
    const long long ntiles = (n_edges + FT_TM - 1) / FT_TM;
    const int chunks = (n_filters + FT_FCH - 1) / FT_FCH;
    const long long want = 1024 / chunks > 1 ? 1024 / chunks : 1;
    dim3 grid((unsigned)(ntiles < want ? ntiles : want), chunks);
    if (GP == 32)
        hipLaunchKernelGGL(cfconv_filter_bf16_kernel<32>, grid, dim3(256), lds, (hipStream_t)stream, d,
                           (long long)n_edges, mu, width, n_gauss, W1, b1, W2, b2, n_filters, out);
    else
        hipLaunchKernelGGL(cfconv_filter_bf16_kernel<64>, grid, dim3(256), lds, (hipStream_t)stream, d,
                           (long long)n_edges, mu, width, n_gauss, W1, b1, W2, b2, n_filters, out);
    MDG_CHECK_LAUNCH("cfconv_filter_bf16_kernel");
    return MDG_OK;
}
