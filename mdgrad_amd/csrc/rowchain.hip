// K13: a CHAIN of node-level Dense layers in one launch.
//
// Between two continuous-filter convolutions everything a SchNet block does is local to one atom's feature row
// (nff/nn/modules.py:543-547 update MLP, nff/nn/models/schnet.py:149-151 residual, the next block's message_node_filter,
// the readout nff/nn/modules.py:761-809) -- and so is the stretch of the reverse sweeps from the readout head back to the
// adjoint of the last aggregation (mdgrad_amd/nn/analytic.py).  csrc/dense.hip runs every one of these layers as its own
// launch; on a single 4 096-bead system that is ~40 launches of ~10 us per MD step, each doing ~1 us of matrix work.  Here
// a workgroup owns 16 rows and walks the whole list of stages itself:
//
//     z   = x B (+ bias)                       B = W^T (Linear layout W[m][k]) or W (trans: the reverse sweeps)
//     act : out0 = ssp(z0), sig = sigmoid(z0), out1 = sig z1            (as mdg_dense)
//     mode: MUL      out0 *= aux0[row, m]
//           HEAD     (after act) pre0 = out0, pre1 = out1 kept ; out0 = sig l_m ; out1 = (1 - sig) pre1 l_m     (l = aux0[m])
//           SSP_BWD  out0 = s z0 ; out1 = (1 - s) td z0 + s z1          (s = aux0[row, m], td = aux1[row, m])
//     then out0 += res0, out1 += res1
//
// The stage's outputs stay in LDS as the next stage's input (and are copied to global memory where the caller wants them:
// the saved activations of the reverse sweep and of the parameter-gradient reductions).  Waves split the OUTPUT COLUMNS
// (tile t = wave + 4 tt), so a workgroup reads every weight exactly once, as B fragments straight from L2 -- 16-byte
// loads along k for the Linear layout with the k permutation dense.hip uses for its A operand (k-step 4 q + c <-> k = 16 q
// + 4 lk + c).  v_mfma_f32_16x16x4_f32, exact f32.  A value another stage of the same launch wrote to global memory
// (sig / t_dot of the update MLP, read back by SSP_BWD) is read by the thread that wrote it: same (row, column) owner.
#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr float LOG2E_C = 1.4426950408889634f;
constexpr float LN2_C = 0.69314718055994531f;
constexpr int RC_ROWS = 16;

struct ChainArgs {
    MdgChainStage s[MDG_CHAIN_MAX_STAGES];
    int n_stages, N, ldt;
};

template <bool DUAL, int TPW>          // TPW: column tiles per wave (2: layers up to 128 wide, 8: up to 512)
__global__ __launch_bounds__(256) void row_chain_kernel(const ChainArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ldt = A.ldt, N = A.N;
    float* X0 = sm;
    float* X1 = sm + RC_ROWS * ldt;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int row0 = blockIdx.x * RC_ROWS;
    for (int si = 0; si < A.n_stages; ++si) {
        const MdgChainStage& S = A.s[si];
        const int K = S.K, M = S.M;
        if (S.in0) {
            // rows of this workgroup from global memory (zero beyond N and beyond K up to a multiple of 4)
            const int K4 = (K + 3) & ~3;
            if ((K & 3) == 0 && (((uintptr_t)S.in0 | (uintptr_t)S.in1) & 15) == 0) {
                const int kq = K >> 2;
                for (int t = tid; t < RC_ROWS * kq; t += 256) {
                    const int r = t / kq, k = (t % kq) * 4, row = row0 + r;
                    float4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
                    if (row < N) {
                        v0 = *reinterpret_cast<const float4*>(S.in0 + (size_t)row * K + k);
                        if (DUAL && S.in1) v1 = *reinterpret_cast<const float4*>(S.in1 + (size_t)row * K + k);
                    }
                    *reinterpret_cast<float4*>(X0 + r * ldt + k) = v0;
                    if (DUAL) *reinterpret_cast<float4*>(X1 + r * ldt + k) = v1;
                }
            } else {
                for (int t = tid; t < RC_ROWS * K4; t += 256) {
                    const int r = t / K4, k = t % K4, row = row0 + r;
                    const bool ok = row < N && k < K;
                    X0[r * ldt + k] = ok ? S.in0[(size_t)row * K + k] : 0.f;
                    if (DUAL) X1[r * ldt + k] = (ok && S.in1) ? S.in1[(size_t)row * K + k] : 0.f;
                }
            }
            __syncthreads();
        }
        const int ntile = (M + 15) >> 4;
        const bool vecb = !S.trans && (K & 3) == 0 && ((uintptr_t)S.W & 15) == 0;
        f32x4 acc0[TPW], acc1[DUAL ? TPW : 1];
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            acc0[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (DUAL) acc1[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int kc = 0; kc < K; kc += 64) {
            float a0[16], a1[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k0 = kc + 16 * q + 4 * lk;
                float4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
                if (k0 < K) {
                    v0 = *reinterpret_cast<const float4*>(X0 + li * ldt + k0);
                    if (DUAL) v1 = *reinterpret_cast<const float4*>(X1 + li * ldt + k0);
                }
                a0[4 * q] = v0.x; a0[4 * q + 1] = v0.y; a0[4 * q + 2] = v0.z; a0[4 * q + 3] = v0.w;
                if (DUAL) { a1[4 * q] = v1.x; a1[4 * q + 1] = v1.y; a1[4 * q + 2] = v1.z; a1[4 * q + 3] = v1.w; }
            }
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt) {
                const int t = wid + 4 * tt;
                if (t >= ntile) continue;
                const int m = t * 16 + li;
                float b[16];
                if (vecb) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int k0 = kc + 16 * q + 4 * lk;
                        float4 v = {0.f, 0.f, 0.f, 0.f};
                        if (m < M && k0 < K) v = *reinterpret_cast<const float4*>(S.W + (size_t)m * K + k0);
                        b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int k = kc + 16 * q + 4 * lk + c;
                            float v = 0.f;
                            if (m < M && k < K) v = S.trans ? S.W[(size_t)k * M + m] : S.W[(size_t)m * K + k];
                            b[4 * q + c] = v;
                        }
                }
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    acc0[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[ks], b[ks], acc0[tt], 0, 0, 0);
                    if (DUAL) acc1[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[ks], b[ks], acc1[tt], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                             // every wave has read its A operand: X may be overwritten
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            const int t = wid + 4 * tt;
            if (t >= ntile) continue;
            const int m = t * 16 + li;
            const bool mok = m < M;
            const float bv = (mok && S.bias) ? S.bias[m] : 0.f;
            const float lv = (mok && S.mode == MDG_CHAIN_HEAD) ? S.aux0[m] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = 4 * lk + r, row = row0 + rr;
                const bool ok = mok && row < N;
                const size_t o = (size_t)row * M + m;
                float z0 = acc0[tt][r] + bv, z1 = DUAL ? acc1[tt][r] : 0.f, sg = 0.f;
                if (S.act == 1) {
                    const float ex = __builtin_amdgcn_exp2f(z0 * LOG2E_C);
                    const bool big = z0 > 20.f;
                    const float sp = __builtin_amdgcn_logf(1.0f + ex) * LN2_C;
                    sg = big ? 1.0f : ex * __builtin_amdgcn_rcpf(1.0f + ex);
                    z0 = (big ? z0 : sp) - LN2_C;
                    z1 *= sg;
                    if (ok && S.sig) S.sig[o] = sg;
                }
                if (S.mode == MDG_CHAIN_MUL) {
                    if (ok) z0 *= S.aux0[o];
                } else if (S.mode == MDG_CHAIN_HEAD) {
                    if (ok && S.pre0) S.pre0[o] = z0;
                    if (ok && DUAL && S.pre1) S.pre1[o] = z1;
                    z0 = sg * lv;
                    z1 = (1.f - sg) * z1 * lv;
                } else if (S.mode == MDG_CHAIN_SSP_BWD) {
                    const float s = ok ? S.aux0[o] : 0.f, td = (ok && DUAL) ? S.aux1[o] : 0.f;
                    const float n0 = s * z0;
                    z1 = (1.f - s) * td * z0 + s * z1;
                    z0 = n0;
                }
                if (ok && S.res0) z0 += S.res0[o];
                if (ok && DUAL && S.res1) z1 += S.res1[o];
                if (!ok) { z0 = 0.f; z1 = 0.f; }
                if (ok && S.out0) S.out0[o] = z0;
                if (ok && DUAL && S.out1) S.out1[o] = z1;
                X0[rr * ldt + m] = z0;
                if (DUAL) X1[rr * ldt + m] = z1;
            }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int mdg_row_chain(const MdgChainStage* stages, int n_stages, int n_rows, int dual, void* stream) {
    MDG_CHECK_ARG(stages && n_stages >= 1 && n_stages <= MDG_CHAIN_MAX_STAGES, "row_chain: 1..%d stages", MDG_CHAIN_MAX_STAGES);
    MDG_CHECK_ARG(n_rows >= 0, "row_chain: bad row count");
    if (n_rows == 0) return MDG_OK;
    ChainArgs a{};
    int wmax = 0, tiles = 0;
    for (int i = 0; i < n_stages; ++i) {
        const MdgChainStage& s = stages[i];
        MDG_CHECK_ARG(s.W && s.K > 0 && s.M > 0 && s.K <= MDG_CHAIN_MAX_WIDTH && s.M <= MDG_CHAIN_MAX_WIDTH,
                      "row_chain: stage %d: null weight or width outside 1..%d", i, MDG_CHAIN_MAX_WIDTH);
        MDG_CHECK_ARG(i > 0 ? (s.in0 || s.K == stages[i - 1].M) : s.in0 != nullptr,
                      "row_chain: stage %d: no input (first stage) or k != the previous stage's width", i);
        MDG_CHECK_ARG(s.act == 0 || s.act == 1, "row_chain: stage %d: act must be 0 or 1", i);
        MDG_CHECK_ARG(s.mode >= MDG_CHAIN_NONE && s.mode <= MDG_CHAIN_SSP_BWD, "row_chain: stage %d: unknown mode", i);
        MDG_CHECK_ARG(s.mode != MDG_CHAIN_HEAD || (s.act == 1 && s.aux0), "row_chain: stage %d: HEAD needs act = 1 and aux0", i);
        MDG_CHECK_ARG(s.mode != MDG_CHAIN_MUL || s.aux0, "row_chain: stage %d: MUL needs aux0", i);
        MDG_CHECK_ARG(s.mode != MDG_CHAIN_SSP_BWD || (s.aux0 && (!dual || s.aux1)), "row_chain: stage %d: SSP_BWD needs aux0 (and aux1)", i);
        wmax = s.K > wmax ? s.K : wmax;
        wmax = s.M > wmax ? s.M : wmax;
        const int t = ((s.M + 15) / 16 + 3) / 4;
        tiles = t > tiles ? t : tiles;
        a.s[i] = s;
    }
    a.n_stages = n_stages; a.N = n_rows;
    a.ldt = ((wmax + 15) & ~15) + 4;                                  // (rows 16 bytes apart modulo the bank cycle)
    const size_t lds = sizeof(float) * RC_ROWS * a.ldt * (dual ? 2 : 1);
    dim3 grid((n_rows + RC_ROWS - 1) / RC_ROWS), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dual) {
        if (tiles <= 2) hipLaunchKernelGGL((row_chain_kernel<true, 2>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((row_chain_kernel<true, 8>), grid, block, lds, st, a);
    } else {
        if (tiles <= 2) hipLaunchKernelGGL((row_chain_kernel<false, 2>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((row_chain_kernel<false, 8>), grid, block, lds, st, a);
    }
    MDG_CHECK_LAUNCH("row_chain_kernel");
    return MDG_OK;
}
