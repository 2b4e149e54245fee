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
#include <stdlib.h>
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

// Weight fragments of column tile t, k chunk kc of a stage: lane (li, lk) gets B[k = kc + 16 q + 4 lk + c][m = 16 t + li] in
// b[4 q + c] (the k order of the activation operand).  Linear layout: four 16-byte loads along k; transposed: 16 loads,
// each coalesced over li.
__device__ __forceinline__ void load_b(const MdgChainStage& S, int kc, int t, int li, int lk, float (&b)[16]) {
    const int K = S.K, M = S.M, m = t * 16 + li;
    const bool vecb = !S.trans && (K & 3) == 0 && ((uintptr_t)S.W & 15) == 0;
    if (vecb) {
        const float* w = S.W + (unsigned)(m * K + kc + 4 * lk);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < M && kc + 16 * q + 4 * lk < K) v = *reinterpret_cast<const float4*>(w + 16 * q);
            b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = kc + 16 * q + 4 * lk + c;
                float v = 0.f;
                if (m < M && k < K) v = S.trans ? S.W[(unsigned)(k * M + m)] : S.W[(unsigned)(m * K + k)];
                b[4 * q + c] = v;
            }
    }
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// two f32 -> two bf16, round to nearest even (v_cvt_pk_bf16_f32): lo half = a
typedef float rc_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 rc_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int rc_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int rc_pk_bf16(float a, float b) {
    const rc_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, rc_bf16x2));
}
__device__ __forceinline__ void st4(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }

// operands of the epilogue for the 4 consecutive columns a lane owns in a tile
struct EpiOps { float4 b, l, x0, x1, r0, r1; };

template <bool DUAL>
__device__ __forceinline__ void epi_load(const MdgChainStage& S, bool mok, bool ok, unsigned m, unsigned o, EpiOps& e) {
    const float4 z = {0.f, 0.f, 0.f, 0.f};
    e.b = (mok && S.bias) ? ld4(S.bias + m) : z;
    e.l = (mok && S.mode == MDG_CHAIN_HEAD) ? ld4(S.aux0 + m) : z;
    const bool ax = ok && (S.mode == MDG_CHAIN_MUL || S.mode == MDG_CHAIN_SSP_BWD);
    e.x0 = ax ? ld4(S.aux0 + o) : z;
    e.x1 = (ax && DUAL && S.mode == MDG_CHAIN_SSP_BWD) ? ld4(S.aux1 + o) : z;
    e.r0 = (ok && S.res0) ? ld4(S.res0 + o) : z;
    e.r1 = (ok && DUAL && S.res1) ? ld4(S.res1 + o) : z;
}

// act / mode / residual on one element; the global copies of sig / pre are written by the caller
template <bool DUAL>
__device__ __forceinline__ void epi_math(int act, int mode, float bv, float lv, float x0, float x1, float q0, float q1, float& z0,
                                         float& z1, float& sg, float& p0, float& p1) {
    z0 += bv;
    sg = 0.f;
    if (act == 1) {
        const float ex = __builtin_amdgcn_exp2f(z0 * LOG2E_C);
        const bool big = z0 > 20.f;
        const float sp = __builtin_amdgcn_logf(1.0f + ex) * LN2_C;
        sg = big ? 1.0f : ex * __builtin_amdgcn_rcpf(1.0f + ex);
        z0 = (big ? z0 : sp) - LN2_C;
        z1 *= sg;
    }
    p0 = z0; p1 = z1;
    if (mode == MDG_CHAIN_MUL) {
        z0 *= x0;
    } else if (mode == MDG_CHAIN_HEAD) {
        z0 = sg * lv;
        z1 = (1.f - sg) * z1 * lv;
    } else if (mode == MDG_CHAIN_SSP_BWD) {
        const float n0 = x0 * z0;
        z1 = (1.f - x0) * x1 * z0 + x0 * z1;
        z0 = n0;
    }
    z0 += q0;
    z1 += q1;
}

// The MFMA takes the WEIGHT fragment as its first operand and the activations as its second: the accumulator of lane
// (li, lk) then holds columns 16 t + 4 lk + [0, 4) of row li -- 16 consecutive bytes of every row-major buffer the epilogue
// touches.  With one wave per SIMD the launch is bound by the instructions a wave issues, not by the matrix pipe: the
// epilogue moves 16-byte vectors (widths that are multiples of 4, aligned buffers; anything else takes the element-wise
// path), offsets are 32-bit, and the operands of the epilogue are requested before the matrix loop.
// LAT: the few-workgroup variant (<= 2 workgroups per CU anyway): epilogue operands requested before the matrix loop, registers
// spent freely.  !LAT: many rows -- what counts is how many workgroups a CU holds, so the operand prefetch goes and the
// register budget is that of three (dual) or four waves per SIMD, no spills (32 768 rows: 6-stage dual chain 114 -> 89 us).
template <bool DUAL, int TPW, bool LAT>          // TPW: column tiles per wave (2: layers up to 128 wide, 8: up to 512)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LAT ? 1 : (TPW <= 2 ? (DUAL ? 3 : 4) : 2))))
void row_chain_kernel(const ChainArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ldt = A.ldt, N = A.N;
    float* X0 = sm;
    float* X1 = sm + RC_ROWS * ldt;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int row0 = blockIdx.x * RC_ROWS;
    const int row = row0 + li;                                       // the row whose outputs this lane owns
    constexpr bool PRE = LAT && TPW <= 2;                            // epilogue operands requested before the matrix loop
    float bc[16];
    if (wid < ((A.s[0].M + 15) >> 4)) load_b(A.s[0], 0, wid, li, lk, bc);
    for (int si = 0; si < A.n_stages; ++si) {
        const MdgChainStage& S = A.s[si];
        const int K = S.K, M = S.M;
        if (S.in0) {
            // rows of this workgroup from global memory (zero beyond N and beyond K up to a multiple of 4)
            const int K4 = (K + 3) & ~3;
            if ((K & 3) == 0 && (((uintptr_t)S.in0 | (uintptr_t)S.in1) & 15) == 0) {
                const int kq = K >> 2;
                for (int t = tid; t < RC_ROWS * kq; t += 256) {
                    const int r = t / kq, k = (t % kq) * 4, rw = row0 + r;
                    float4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
                    if (rw < N) {
                        v0 = ld4(S.in0 + (unsigned)(rw * K + k));
                        if (DUAL && S.in1) v1 = ld4(S.in1 + (unsigned)(rw * K + k));
                    }
                    *reinterpret_cast<float4*>(X0 + r * ldt + k) = v0;
                    if (DUAL) *reinterpret_cast<float4*>(X1 + r * ldt + k) = v1;
                }
            } else {
                for (int t = tid; t < RC_ROWS * K4; t += 256) {
                    const int r = t / K4, k = t % K4, rw = row0 + r;
                    const bool ok = rw < N && k < K;
                    X0[r * ldt + k] = ok ? S.in0[(unsigned)(rw * K + k)] : 0.f;
                    if (DUAL) X1[r * ldt + k] = (ok && S.in1) ? S.in1[(unsigned)(rw * K + k)] : 0.f;
                }
            }
            __syncthreads();
        }
        const int ntile = (M + 15) >> 4;
        const int ntt = wid < ntile ? (ntile - wid + 3) >> 2 : 0;     // tiles of this wave: t = wid + 4 tt, tt < ntt
        const uintptr_t al = (uintptr_t)S.bias | (uintptr_t)S.aux0 | (uintptr_t)S.aux1 | (uintptr_t)S.res0 | (uintptr_t)S.res1 |
                             (uintptr_t)S.out0 | (uintptr_t)S.out1 | (uintptr_t)S.sig | (uintptr_t)S.pre0 | (uintptr_t)S.pre1;
        const bool fast = (M & 3) == 0 && (al & 15) == 0;
        EpiOps eo[PRE ? TPW : 1];
        if constexpr (PRE) {
            if (fast) {
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt) {
                    const int m = (wid + 4 * tt) * 16 + 4 * lk;
                    const bool mok = tt < ntt && m < M;
                    epi_load<DUAL>(S, mok, mok && row < N, m, row * M + m, eo[tt]);
                }
            }
        }
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
                if (tt >= ntt) continue;
                // the fragments of the next step of this wave: next tile of this chunk, first tile of the next chunk, or
                // the first step of the next stage
                float bn[16];
                if (tt + 1 < ntt) load_b(S, kc, wid + 4 * (tt + 1), li, lk, bn);
                else if (kc + 64 < K) load_b(S, kc + 64, wid, li, lk, bn);
                else if (si + 1 < A.n_stages && wid < ((A.s[si + 1].M + 15) >> 4)) load_b(A.s[si + 1], 0, wid, li, lk, bn);
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    acc0[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bc[ks], a0[ks], acc0[tt], 0, 0, 0);
                    if (DUAL) acc1[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bc[ks], a1[ks], acc1[tt], 0, 0, 0);
                }
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) bc[ks] = bn[ks];
            }
        }
        // (a wave without a tile in this stage still has to fetch its first fragments of the next one)
        if (ntt == 0 && si + 1 < A.n_stages && wid < ((A.s[si + 1].M + 15) >> 4)) load_b(A.s[si + 1], 0, wid, li, lk, bc);
        __syncthreads();                                             // every wave has read its activations: X may be overwritten
        const int act = S.act, mode = S.mode;
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            if (tt >= ntt) continue;
            const int m = (wid + 4 * tt) * 16 + 4 * lk;               // first of the lane's 4 columns
            if (fast) {
                const bool mok = m < M, ok = mok && row < N;
                const unsigned o = row * M + m;
                EpiOps e;
                if constexpr (PRE) e = eo[tt];
                else epi_load<DUAL>(S, mok, ok, m, o, e);
                const float bv[4] = {e.b.x, e.b.y, e.b.z, e.b.w}, lv[4] = {e.l.x, e.l.y, e.l.z, e.l.w};
                const float x0[4] = {e.x0.x, e.x0.y, e.x0.z, e.x0.w}, x1[4] = {e.x1.x, e.x1.y, e.x1.z, e.x1.w};
                const float q0[4] = {e.r0.x, e.r0.y, e.r0.z, e.r0.w}, q1[4] = {e.r1.x, e.r1.y, e.r1.z, e.r1.w};
                float z0[4], z1[4], sg[4], p0[4], p1[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    z0[r] = acc0[tt][r]; z1[r] = DUAL ? acc1[tt][r] : 0.f;
                    epi_math<DUAL>(act, mode, bv[r], lv[r], x0[r], x1[r], q0[r], q1[r], z0[r], z1[r], sg[r], p0[r], p1[r]);
                    if (!ok) { z0[r] = 0.f; z1[r] = 0.f; }
                }
                if (ok) {
                    if (act == 1 && S.sig) st4(S.sig + o, sg);
                    if (mode == MDG_CHAIN_HEAD) {
                        if (S.pre0) st4(S.pre0 + o, p0);
                        if (DUAL && S.pre1) st4(S.pre1 + o, p1);
                    }
                    if (S.out0) st4(S.out0 + o, z0);
                    if (DUAL && S.out1) st4(S.out1 + o, z1);
                    if (S.out0_h) *reinterpret_cast<rc_u32x2*>(S.out0_h + o) = rc_u32x2{rc_pk_bf16(z0[0], z0[1]), rc_pk_bf16(z0[2], z0[3])};
                    if (DUAL && S.out1_h) *reinterpret_cast<rc_u32x2*>(S.out1_h + o) = rc_u32x2{rc_pk_bf16(z1[0], z1[1]), rc_pk_bf16(z1[2], z1[3])};
                }
                if (mok) {
                    st4(X0 + li * ldt + m, z0);
                    if (DUAL) st4(X1 + li * ldt + m, z1);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mm = m + r;
                    const bool mok = mm < M, ok = mok && row < N;
                    const unsigned o = row * M + mm;
                    const float bv = (mok && S.bias) ? S.bias[mm] : 0.f;
                    const float lv = (mok && mode == MDG_CHAIN_HEAD) ? S.aux0[mm] : 0.f;
                    const bool ax = ok && (mode == MDG_CHAIN_MUL || mode == MDG_CHAIN_SSP_BWD);
                    const float x0 = ax ? S.aux0[o] : 0.f, x1 = (ax && DUAL && mode == MDG_CHAIN_SSP_BWD) ? S.aux1[o] : 0.f;
                    const float q0 = (ok && S.res0) ? S.res0[o] : 0.f, q1 = (ok && DUAL && S.res1) ? S.res1[o] : 0.f;
                    float z0 = acc0[tt][r], z1 = DUAL ? acc1[tt][r] : 0.f, sg, p0, p1;
                    epi_math<DUAL>(act, mode, bv, lv, x0, x1, q0, q1, z0, z1, sg, p0, p1);
                    if (!ok) { z0 = 0.f; z1 = 0.f; }
                    if (ok) {
                        if (act == 1 && S.sig) S.sig[o] = sg;
                        if (mode == MDG_CHAIN_HEAD) {
                            if (S.pre0) S.pre0[o] = p0;
                            if (DUAL && S.pre1) S.pre1[o] = p1;
                        }
                        if (S.out0) S.out0[o] = z0;
                        if (DUAL && S.out1) S.out1[o] = z1;
                        if (S.out0_h) S.out0_h[o] = (uint16_t)rc_pk_bf16(z0, 0.f);
                        if (DUAL && S.out1_h) S.out1_h[o] = (uint16_t)rc_pk_bf16(z1, 0.f);
                    }
                    X0[li * ldt + mm] = z0;                           // (columns [M, 16 ntile) are left at zero)
                    if (DUAL) X1[li * ldt + mm] = z1;
                }
            }
        }
        __syncthreads();
    }
}

// ======================================================================================= specialised chains
// The walker above reads a stage's shape, flags and pointers at run time; with one wave per SIMD that interpretation is
// most of what a stage costs (a chain of plain 64 -> 64 stages: 2.15 us per stage through the walker, 0.32 us with
// everything known at compile time, same MFMAs).  The three stretches the SchNet sweeps consist of are therefore ALSO
// compiled with their shapes and epilogues fixed -- mdg_row_chain recognises them in the descriptor list and takes the
// compiled version when the widths are among the instantiated ones -- (A, F) in {64, 128}^2, the lower half of the reference's
// search space (demo/fit_rdf_gnn.py:16-19); every other list goes through the walker.
//   FWD    [F -> A, ssp] [A -> A, + residual] [A -> F]                                   update MLP, residual, next node filter
//   TURN   [F -> A, ssp] [A -> A, + residual] [A -> A/2, ssp, head] [A/2 -> A]^T [A -> A]^T ssp' [A -> F]^T
//   REV    [F -> A]^T + residual, [A -> A]^T ssp', [A -> F]^T
// All weight fragments of the launch (<= 120 registers per lane for TURN at A = 64, F = 128) are requested at kernel start,
// in one round trip, together with the first input rows; sigmoid / tangent rows a later stage of the same launch needs stay
// in registers (TURN).
template <int M> constexpr int spec_tp() { return (M / 16 + 3) / 4; }          // column tiles per wave

__device__ __forceinline__ void st4v(float* p, unsigned o, const float (&v)[4]) {
    if (p) *reinterpret_cast<float4*>(p + o) = make_float4(v[0], v[1], v[2], v[3]);
}
// the bf16 mirror of an output row piece (MdgChainStage::out0_h / out1_h): what the rows16 cfconv kernels gather
__device__ __forceinline__ void st4h(uint16_t* p, unsigned o, const float (&v)[4]) {
    if (p) *reinterpret_cast<rc_u32x2*>(p + o) = rc_u32x2{rc_pk_bf16(v[0], v[1]), rc_pk_bf16(v[2], v[3])};
}
__device__ __forceinline__ void ld4v(const float* p, unsigned o, bool ok, float (&v)[4]) {
    float4 q = {0.f, 0.f, 0.f, 0.f};
    if (ok && p) q = ld4(p + o);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}

// X3 (round 6): the stage products as THREE bf16 MFMAs on operands split into a bf16 head and a bf16 remainder,
//     x w  ~  x_h w_h + x_h w_l + x_l w_h                          (dropped: x_l w_l, 2^-18 of the product; heads round to nearest)
// -- 3 v_mfma_f32_16x16x32_bf16 (3 x 16 cycles for 32 k) where the f32 form issues 8 v_mfma_f32_16x16x4_f32 (8 x 32 cycles): the
// matrix time of a chain drops to 3/16.  The launches of the stacked trajectories are bound by exactly that time (dual forward
// chain at 32 768 rows: 39.8 us, 26 us with the matrix instructions cut to a quarter).  ~1e-5 relative per product against
// 6e-8: taken only where the caller says so (flag MDG_CHAIN_X3: the rows16 precision option, whose gathered rows are bf16).
// LDS holds the stage inputs as two bf16 planes [16][ldt] (head, remainder) per row set; accumulation and epilogues are f32.
typedef short rc_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int rc_u32x4 __attribute__((ext_vector_type(4)));
// X6 (round 6, the f32 chains): THREE bf16 pieces per operand (8 + 8 + 8 bits: they add up to the f32 value exactly) and the six
// piece products that matter, smallest first -- error 1.8e-7 of sum |terms| against 1.9e-7 for v_mfma_f32_16x16x4_f32
// (tools/micro/split_mfma.hip) at 6 x 16 instead of 8 x 32 cycles per 32 k: f32 accuracy on the bf16 matrix pipe.  Flag
// MDG_CHAIN_X6; the compiled chains honour it.  NP below: planes per operand (0: f32 matrix instruction, 2: X3, 3: X6).
template <int NP, int K, int M> struct WFrag { rc_bf16x8 p[NP][spec_tp<M>()][K / 32]; };
template <int K, int M> struct WFrag<0, K, M> { float b[spec_tp<M>()][K / 4]; };

// NP pieces of two floats, packed (low half = a): head, (middle,) remainder -- each the bf16 nearest to what is left
template <int NP>
__device__ __forceinline__ void rc_split2(float a, float b, unsigned int (&u)[NP]) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        u[p] = rc_pk_bf16(a, b);
        a -= __uint_as_float(u[p] << 16);
        b -= __uint_as_float(u[p] & 0xffff0000u);
    }
}
template <int NP>
__device__ __forceinline__ void rc_split8(const float (&f)[8], rc_bf16x8 (&out)[NP]) {
    rc_u32x4 v[NP];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned int u[NP];
        rc_split2<NP>(f[2 * i], f[2 * i + 1], u);
#pragma unroll
        for (int p = 0; p < NP; ++p) v[p][i] = u[p];
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) out[p] = __builtin_bit_cast(rc_bf16x8, v[p]);
}

// lane (li, lk): W_eff[m = 16 t + li][k = 32 kc + 8 lk + 0..7], split
template <int K, int M, bool TRANS, int NP>
__device__ __forceinline__ void spec_load_w(const float* __restrict__ W, int wid, int li, int lk, WFrag<NP, K, M>& w) {
#pragma unroll
    for (int tt = 0; tt < spec_tp<M>(); ++tt) {
        const int t = wid + 4 * tt;
        if (t * 16 >= M) continue;
        const int m = t * 16 + li;
#pragma unroll
        for (int kc = 0; kc < K / 32; ++kc) {
            float f[8];
            if constexpr (!TRANS) {
                const float4 u = ld4(W + m * K + 32 * kc + 8 * lk), v = ld4(W + m * K + 32 * kc + 8 * lk + 4);
                f[0] = u.x; f[1] = u.y; f[2] = u.z; f[3] = u.w; f[4] = v.x; f[5] = v.y; f[6] = v.z; f[7] = v.w;
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) f[c] = W[(32 * kc + 8 * lk + c) * M + m];
            }
            rc_bf16x8 pc[NP];
            rc_split8<NP>(f, pc);
#pragma unroll
            for (int p = 0; p < NP; ++p) w.p[p][tt][kc] = pc[p];
        }
    }
}

// the split image of a row set: NP planes [RC_ROWS][ldt] bf16 (ldt elements = 2 ldt bytes per row), plane 0 = heads
template <int NP>
__device__ __forceinline__ void rc_put4(float* X, int ldt, int r, int k, float a, float b, float c, float d) {
    unsigned short* P = reinterpret_cast<unsigned short*>(X);
    unsigned int u0[NP], u1[NP];
    rc_split2<NP>(a, b, u0);
    rc_split2<NP>(c, d, u1);
#pragma unroll
    for (int p = 0; p < NP; ++p) *reinterpret_cast<rc_u32x2*>(P + (p * RC_ROWS + r) * ldt + k) = rc_u32x2{u0[p], u1[p]};
}

// piece products, smallest first: (weight piece, row piece) with piece index sum <= NP - 1
template <bool DUAL, int K, int M, int NP>
__device__ __forceinline__ void spec_mma(const float* X0, const float* X1, int ldt, int wid, int li, int lk, const WFrag<NP, K, M>& w,
                                         f32x4 (&acc0)[spec_tp<M>()], f32x4 (&acc1)[spec_tp<M>()]) {
    const unsigned short* P0 = reinterpret_cast<const unsigned short*>(X0);
    const unsigned short* P1 = reinterpret_cast<const unsigned short*>(X1);
#pragma unroll
    for (int tt = 0; tt < spec_tp<M>(); ++tt) { acc0[tt] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[tt] = acc0[tt]; }
#pragma unroll
    for (int kc = 0; kc < K / 32; ++kc) {
        const int o = li * ldt + 32 * kc + 8 * lk;
        rc_bf16x8 x[NP], y[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            x[p] = *reinterpret_cast<const rc_bf16x8*>(P0 + p * RC_ROWS * ldt + o);
            y[p] = x[p];
            if (DUAL) y[p] = *reinterpret_cast<const rc_bf16x8*>(P1 + p * RC_ROWS * ldt + o);
        }
#pragma unroll
        for (int tt = 0; tt < spec_tp<M>(); ++tt) {
            if ((wid + 4 * tt) * 16 >= M) continue;
#pragma unroll
            for (int lvl = NP - 1; lvl >= 0; --lvl)
#pragma unroll
                for (int pw = lvl; pw >= 0; --pw) {
                    const int px = lvl - pw;
                    acc0[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p[pw][tt][kc], x[px], acc0[tt], 0, 0, 0);
                    if (DUAL) acc1[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p[pw][tt][kc], y[px], acc1[tt], 0, 0, 0);
                }
        }
    }
}

template <int K, int M, bool TRANS>
__device__ __forceinline__ void spec_load_w(const float* __restrict__ W, int wid, int li, int lk, WFrag<0, K, M>& w) {
    float (&b)[spec_tp<M>()][K / 4] = w.b;
#pragma unroll
    for (int tt = 0; tt < spec_tp<M>(); ++tt) {
        const int t = wid + 4 * tt;
        if (t * 16 >= M) continue;
        const int m = t * 16 + li;
#pragma unroll
        for (int q = 0; q < K / 16; ++q) {
            if constexpr (!TRANS) {
                const float4 v = ld4(W + m * K + 16 * q + 4 * lk);
                b[tt][4 * q] = v.x; b[tt][4 * q + 1] = v.y; b[tt][4 * q + 2] = v.z; b[tt][4 * q + 3] = v.w;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) b[tt][4 * q + c] = W[(16 * q + 4 * lk + c) * M + m];
            }
        }
    }
}

// bias (and the head vector of a HEAD stage) of the lane's 4 columns per tile.  The forward chain requests them BEFORE the
// stage's matrix products, so that the epilogue behind the barrier does not start with a round trip of its own (dual forward
// chain at 32 768 rows: 45.1 -> 41.9 us); the turn / reverse chains -- more weights in flight, no biases on their transposed
// stages -- measured no gain from the same reordering and keep the loads where the values are used.
template <int M, int MODE>
__device__ __forceinline__ void spec_load_bias(const MdgChainStage& S, int wid, int lk, float (&bv)[spec_tp<M>()][4], float (&lv)[spec_tp<M>()][4]) {
#pragma unroll
    for (int tt = 0; tt < spec_tp<M>(); ++tt) {
        const int m = (wid + 4 * tt) * 16 + 4 * lk;
        const bool on = (wid + 4 * tt) * 16 < M;
        ld4v(S.bias, m, on, bv[tt]);
        ld4v(MODE == MDG_CHAIN_HEAD ? S.aux0 : nullptr, m, on, lv[tt]);
    }
}

template <bool DUAL, int K, int NP = 0>
__device__ __forceinline__ void spec_load_x(const float* __restrict__ in0, const float* __restrict__ in1, float* X0, float* X1,
                                            int ldt, int row0, int N, int tid) {
    constexpr int KQ = K / 4;
#pragma unroll
    for (int t0 = 0; t0 < RC_ROWS * KQ; t0 += 256) {
        const int t = t0 + tid;
        if (RC_ROWS * KQ % 256 != 0 && t >= RC_ROWS * KQ) break;
        const int r = t / KQ, k = (t % KQ) * 4, rw = row0 + r;
        float4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (rw < N) {
            v0 = ld4(in0 + (unsigned)(rw * K + k));
            if (DUAL && in1) v1 = ld4(in1 + (unsigned)(rw * K + k));
        }
        if constexpr (NP > 0) {
            rc_put4<NP>(X0, ldt, r, k, v0.x, v0.y, v0.z, v0.w);
            if (DUAL) rc_put4<NP>(X1, ldt, r, k, v1.x, v1.y, v1.z, v1.w);
        } else {
            *reinterpret_cast<float4*>(X0 + r * ldt + k) = v0;
            if (DUAL) *reinterpret_cast<float4*>(X1 + r * ldt + k) = v1;
        }
    }
}

// the same in two halves for the LOOP kernels: the next tile's rows are requested into registers while the current tile is
// worked on, and written to LDS when its turn comes
template <int K> constexpr int spec_xq() { return (RC_ROWS * (K / 4) + 255) / 256; }
template <bool DUAL, int K>
__device__ __forceinline__ void spec_fetch_x(const float* __restrict__ in0, const float* __restrict__ in1, int row0, int N, int tid,
                                             float4 (&v0)[spec_xq<K>()], float4 (&v1)[spec_xq<K>()]) {
    constexpr int KQ = K / 4;
#pragma unroll
    for (int i = 0; i < spec_xq<K>(); ++i) {
        const int t = i * 256 + tid, r = t / KQ, k = (t % KQ) * 4, rw = row0 + r;
        v0[i] = make_float4(0.f, 0.f, 0.f, 0.f); v1[i] = v0[i];
        if (t < RC_ROWS * KQ && rw < N) {
            v0[i] = ld4(in0 + (unsigned)(rw * K + k));
            if (DUAL && in1) v1[i] = ld4(in1 + (unsigned)(rw * K + k));
        }
    }
}
template <bool DUAL, int K, int NP = 0>
__device__ __forceinline__ void spec_put_x(float* X0, float* X1, int ldt, int tid, const float4 (&v0)[spec_xq<K>()],
                                           const float4 (&v1)[spec_xq<K>()]) {
    constexpr int KQ = K / 4;
#pragma unroll
    for (int i = 0; i < spec_xq<K>(); ++i) {
        const int t = i * 256 + tid, r = t / KQ, k = (t % KQ) * 4;
        if (RC_ROWS * KQ % 256 != 0 && t >= RC_ROWS * KQ) break;
        if constexpr (NP > 0) {
            rc_put4<NP>(X0, ldt, r, k, v0[i].x, v0[i].y, v0[i].z, v0[i].w);
            if (DUAL) rc_put4<NP>(X1, ldt, r, k, v1[i].x, v1[i].y, v1[i].z, v1[i].w);
        } else {
            *reinterpret_cast<float4*>(X0 + r * ldt + k) = v0[i];
            if (DUAL) *reinterpret_cast<float4*>(X1 + r * ldt + k) = v1[i];
        }
    }
}
template <int M>
__device__ __forceinline__ void spec_copy(float (&d)[spec_tp<M>()][4], const float (&s)[spec_tp<M>()][4]) {
#pragma unroll
    for (int tt = 0; tt < spec_tp<M>(); ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) d[tt][r] = s[tt][r];
}

// matrix part of a stage: acc[tt] (+)= rows of X (LDS) x the wave's weight fragments; lane (li, lk) ends up with columns
// 16 t + 4 lk + [0, 4) of row li
template <bool DUAL, int K, int M>
__device__ __forceinline__ void spec_mma(const float* X0, const float* X1, int ldt, int wid, int li, int lk,
                                         const WFrag<0, K, M>& w, f32x4 (&acc0)[spec_tp<M>()], f32x4 (&acc1)[spec_tp<M>()]) {
    const float (&b)[spec_tp<M>()][K / 4] = w.b;
#pragma unroll
    for (int tt = 0; tt < spec_tp<M>(); ++tt) { acc0[tt] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[tt] = acc0[tt]; }
#pragma unroll
    for (int q = 0; q < K / 16; ++q) {
        const float4 v0 = *reinterpret_cast<const float4*>(X0 + li * ldt + 16 * q + 4 * lk);
        float4 v1 = v0;
        if (DUAL) v1 = *reinterpret_cast<const float4*>(X1 + li * ldt + 16 * q + 4 * lk);
        const float a0[4] = {v0.x, v0.y, v0.z, v0.w}, a1[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int tt = 0; tt < spec_tp<M>(); ++tt) {
            if ((wid + 4 * tt) * 16 >= M) continue;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#ifdef MDG_CHAIN_FAKE
                if (c) continue;
#endif
                acc0[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[tt][4 * q + c], a0[c], acc0[tt], 0, 0, 0);
                if (DUAL) acc1[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[tt][4 * q + c], a1[c], acc1[tt], 0, 0, 0);
            }
        }
    }
}


// epilogue of a stage on the lane's 4 columns of each of its tiles; x0 / x1: operands of MUL / SSP_BWD (registers), q0 / q1:
// residuals (registers); sg / kd receive sigmoid and the tangent row of an activation stage
template <bool DUAL, int M, int ACT, int MODE, int NP = 0>
__device__ __forceinline__ void spec_epilogue(const MdgChainStage& S, f32x4 (&acc0)[spec_tp<M>()], f32x4 (&acc1)[spec_tp<M>()],
                                              const float (&x0)[spec_tp<M>()][4], const float (&x1)[spec_tp<M>()][4],
                                              const float (&q0)[spec_tp<M>()][4], const float (&q1)[spec_tp<M>()][4],
                                              float (&sgk)[spec_tp<M>()][4], float (&tdk)[spec_tp<M>()][4], float* X0, float* X1, int ldt,
                                              int row, int N, int wid, int li, int lk, const float (&bvs)[spec_tp<M>()][4],
                                              const float (&lvs)[spec_tp<M>()][4]) {
#pragma unroll
    for (int tt = 0; tt < spec_tp<M>(); ++tt) {
        const int m = (wid + 4 * tt) * 16 + 4 * lk;
        if ((wid + 4 * tt) * 16 >= M) continue;
        const bool ok = row < N;
        const unsigned o = (unsigned)(row * M + m);
        float z0[4], z1[4], sg[4], p0[4], p1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            z0[r] = acc0[tt][r]; z1[r] = DUAL ? acc1[tt][r] : 0.f;
            epi_math<DUAL>(ACT, MODE, bvs[tt][r], MODE == MDG_CHAIN_HEAD ? lvs[tt][r] : 0.f, x0[tt][r], x1[tt][r], q0[tt][r], q1[tt][r], z0[r], z1[r],
                           sg[r], p0[r], p1[r]);
            if (ACT == 1) { sgk[tt][r] = sg[r]; tdk[tt][r] = p1[r]; }
            if (!ok) { z0[r] = 0.f; z1[r] = 0.f; }
        }
        if (ok) {
            if (ACT == 1) st4v(S.sig, o, sg);
            if (MODE == MDG_CHAIN_HEAD) { st4v(S.pre0, o, p0); if (DUAL) st4v(S.pre1, o, p1); }
            st4v(S.out0, o, z0);
            if (DUAL) st4v(S.out1, o, z1);
            if (S.out0_h) {                                       // (uniform, rare)
                st4h(S.out0_h, o, z0);
                if (DUAL) st4h(S.out1_h, o, z1);
            }
        }
        if constexpr (NP > 0) {
            rc_put4<NP>(X0, ldt, li, m, z0[0], z0[1], z0[2], z0[3]);
            if (DUAL) rc_put4<NP>(X1, ldt, li, m, z1[0], z1[1], z1[2], z1[3]);
        } else {
            *reinterpret_cast<float4*>(X0 + li * ldt + m) = make_float4(z0[0], z0[1], z0[2], z0[3]);
            if (DUAL) *reinterpret_cast<float4*>(X1 + li * ldt + m) = make_float4(z1[0], z1[1], z1[2], z1[3]);
        }
    }
}

template <int M>
__device__ __forceinline__ void spec_zero(float (&v)[spec_tp<M>()][4]) {
#pragma unroll
    for (int tt = 0; tt < spec_tp<M>(); ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[tt][r] = 0.f;
}

// residual / operand rows of a stage, requested early
template <int M>
__device__ __forceinline__ void spec_load_rows(const float* p, int row, int N, int wid, int lk, float (&v)[spec_tp<M>()][4]) {
#pragma unroll
    for (int tt = 0; tt < spec_tp<M>(); ++tt) {
        const int m = (wid + 4 * tt) * 16 + 4 * lk;
        ld4v(p, (unsigned)(row * M + m), (wid + 4 * tt) * 16 < M && row < N, v[tt]);
    }
}

enum { SPEC_FWD = 0, SPEC_TURN = 1, SPEC_REV = 2 };

// (a workgroup's life is one round trip for its inputs, a few small matrix products between barriers, and its stores: the
//  launch is as fast as there are workgroups in flight -- the register budget is that of four waves per SIMD where it costs no
//  scratch (n_atom_basis = 64): the turn chain 136 -> 116 registers, the forward chain 104 -> 90; the reverse chain would spill
//  two dwords and the 128-wide chains hundreds of bytes: they keep the compiler's own budget)
// LOOP (round 6): many rows -- a stack of replicas, 2 048 row tiles -- as ONE round of workgroups has every workgroup of the
// chip in the same phase at the same time (all loading, then all multiplying, then all storing) and re-reads the chain's
// weights from L2 once per 16 rows (80 KB per workgroup: 164 MB per launch beside ~110 MB of rows).  The LOOP instantiation is
// launched with a few workgroups per CU; each keeps ALL weight fragments and biases of the chain in registers and walks row
// tiles blockIdx.x, + gridDim.x, ...: the weights are read once per workgroup and the workgroups of a CU drift out of phase.
template <int A_, int F_, bool DUAL, int KIND, bool LOOP, int NP>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(NP == 3 ? ((LOOP || A_ == 64) ? 2 : 1) : (LOOP ? 2 : ((A_ == 64 && KIND != 2) ? 4 : 1)))))
void chain_spec_kernel(const ChainArgs A) {
    constexpr int H_ = A_ / 2;                                                      // readout hidden width
    constexpr int WMAX = A_ > F_ ? A_ : F_, ldt = WMAX + (NP ? 8 : 4);       // (NP > 0: bf16 elements per plane row)
    constexpr int XSZ = RC_ROWS * ldt * (NP == 3 ? 3 : 2) / 2;                     // floats of one row set's image
    __shared__ __attribute__((aligned(16))) float Xs[(DUAL ? 2 : 1) * XSZ];
    float* X0 = Xs;
    float* X1 = Xs + (DUAL ? XSZ : 0);
    const int N = A.N, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int n_tiles = (N + RC_ROWS - 1) / RC_ROWS, t_step = LOOP ? (int)gridDim.x : n_tiles;
    float sgA[spec_tp<A_>()][4], tdA[spec_tp<A_>()][4], zA[spec_tp<A_>()][4], zF[spec_tp<F_>()][4], zH[spec_tp<H_>()][4];
    float dF[spec_tp<F_>()][4], dH[spec_tp<H_>()][4];                              // (sigmoid / tangent sinks of stages that keep none)
    spec_zero<A_>(zA); spec_zero<F_>(zF); spec_zero<H_>(zH);
    if constexpr (KIND == SPEC_FWD) {
        // ---- first round trip: the weights of the first two stages, the input rows, the residual rows, the biases.  The
        //      weights of the third stage are requested while the first multiplies, each bias before its stage's products.
        WFrag<NP, F_, A_> w0; WFrag<NP, A_, A_> w1; WFrag<NP, A_, F_> w2;
        float b0[spec_tp<A_>()][4], b1[spec_tp<A_>()][4], lA[spec_tp<A_>()][4], b2[spec_tp<F_>()][4], lF[spec_tp<F_>()][4];
        spec_load_w<F_, A_, false>(A.s[0].W, wid, li, lk, w0);
        spec_load_w<A_, A_, false>(A.s[1].W, wid, li, lk, w1);
        if constexpr (LOOP) {
            spec_load_w<A_, F_, false>(A.s[2].W, wid, li, lk, w2);
            spec_load_bias<A_, MDG_CHAIN_NONE>(A.s[0], wid, lk, b0, lA);
            spec_load_bias<A_, MDG_CHAIN_NONE>(A.s[1], wid, lk, b1, lA);
            spec_load_bias<F_, MDG_CHAIN_NONE>(A.s[2], wid, lk, b2, lF);
        }
        float4 px0[spec_xq<F_>()], px1[spec_xq<F_>()];                            // LOOP: the next tile's rows and residuals
        float pq0[spec_tp<A_>()][4], pq1[spec_tp<A_>()][4];
        if constexpr (LOOP) {
            spec_fetch_x<DUAL, F_>(A.s[0].in0, A.s[0].in1, blockIdx.x * RC_ROWS, N, tid, px0, px1);
            spec_load_rows<A_>(A.s[1].res0, blockIdx.x * RC_ROWS + li, N, wid, lk, pq0);
            spec_load_rows<A_>(DUAL ? A.s[1].res1 : nullptr, blockIdx.x * RC_ROWS + li, N, wid, lk, pq1);
        }
        for (int tile = blockIdx.x, it = 0; LOOP ? tile < n_tiles : it < 1; tile += t_step, ++it) {
            const int row0 = tile * RC_ROWS, row = row0 + li;
            float q0[spec_tp<A_>()][4], q1[spec_tp<A_>()][4];
            if constexpr (LOOP) {
                spec_put_x<DUAL, F_, NP>(X0, X1, ldt, tid, px0, px1);
                spec_copy<A_>(q0, pq0); spec_copy<A_>(q1, pq1);
                const int nrow0 = (tile + t_step) * RC_ROWS;                        // (past the end: nothing is loaded)
                spec_fetch_x<DUAL, F_>(A.s[0].in0, A.s[0].in1, nrow0, N, tid, px0, px1);
                spec_load_rows<A_>(A.s[1].res0, nrow0 + li, N, wid, lk, pq0);
                spec_load_rows<A_>(DUAL ? A.s[1].res1 : nullptr, nrow0 + li, N, wid, lk, pq1);
            } else {
                spec_load_x<DUAL, F_, NP>(A.s[0].in0, A.s[0].in1, X0, X1, ldt, row0, N, tid);
                spec_load_rows<A_>(A.s[1].res0, row, N, wid, lk, q0);
                spec_load_rows<A_>(DUAL ? A.s[1].res1 : nullptr, row, N, wid, lk, q1);
                spec_load_bias<A_, MDG_CHAIN_NONE>(A.s[0], wid, lk, b0, lA);
                spec_load_bias<A_, MDG_CHAIN_NONE>(A.s[1], wid, lk, b1, lA);
            }
            f32x4 a0[spec_tp<A_>()], a1[spec_tp<A_>()];
            __syncthreads();
            // stage 0: t = ssp(U1 m + c1), su, td
            spec_mma<DUAL, F_, A_>(X0, X1, ldt, wid, li, lk, w0, a0, a1);
            if constexpr (!LOOP) {
                spec_load_w<A_, F_, false>(A.s[2].W, wid, li, lk, w2);
                spec_load_bias<F_, MDG_CHAIN_NONE>(A.s[2], wid, lk, b2, lF);
            }
            __syncthreads();
            spec_epilogue<DUAL, A_, 1, MDG_CHAIN_NONE, NP>(A.s[0], a0, a1, zA, zA, zA, zA, sgA, tdA, X0, X1, ldt, row, N, wid, li, lk, b0, zA);
            __syncthreads();
            // stage 1: r' = U2 t + c2 + r
            spec_mma<DUAL, A_, A_>(X0, X1, ldt, wid, li, lk, w1, a0, a1);
            __syncthreads();
            {
                float s_[spec_tp<A_>()][4], t_[spec_tp<A_>()][4];
                spec_epilogue<DUAL, A_, 0, MDG_CHAIN_NONE, NP>(A.s[1], a0, a1, zA, zA, q0, q1, s_, t_, X0, X1, ldt, row, N, wid, li, lk, b1, zA);
            }
            __syncthreads();
            // stage 2: h' = Wn' r' + bn'
            f32x4 f0[spec_tp<F_>()], f1[spec_tp<F_>()];
            spec_mma<DUAL, A_, F_>(X0, X1, ldt, wid, li, lk, w2, f0, f1);
            __syncthreads();
            spec_epilogue<DUAL, F_, 0, MDG_CHAIN_NONE, NP>(A.s[2], f0, f1, zF, zF, zF, zF, dF, dF, X0, X1, ldt, row, N, wid, li, lk, b2, zF);
            if constexpr (LOOP) __syncthreads();                                   // (the next tile's rows overwrite X)
        }
    } else if constexpr (KIND == SPEC_TURN) {
        WFrag<NP, F_, A_> w0; WFrag<NP, A_, A_> w1;
        WFrag<NP, A_, H_> w2; WFrag<NP, H_, A_> w3; WFrag<NP, A_, A_> w4; WFrag<NP, A_, F_> w5;
        float bA0[spec_tp<A_>()][4], bA1[spec_tp<A_>()][4], lA[spec_tp<A_>()][4], bH[spec_tp<H_>()][4], lH[spec_tp<H_>()][4];
        spec_load_w<F_, A_, false>(A.s[0].W, wid, li, lk, w0);
        spec_load_w<A_, A_, false>(A.s[1].W, wid, li, lk, w1);
        if constexpr (LOOP) {
            spec_load_w<A_, H_, false>(A.s[2].W, wid, li, lk, w2);
            spec_load_w<H_, A_, true>(A.s[3].W, wid, li, lk, w3);
            spec_load_w<A_, A_, true>(A.s[4].W, wid, li, lk, w4);
            spec_load_w<A_, F_, true>(A.s[5].W, wid, li, lk, w5);
            spec_load_bias<A_, MDG_CHAIN_NONE>(A.s[0], wid, lk, bA0, lA);
            spec_load_bias<A_, MDG_CHAIN_NONE>(A.s[1], wid, lk, bA1, lA);
            spec_load_bias<H_, MDG_CHAIN_HEAD>(A.s[2], wid, lk, bH, lH);
        }
        float4 px0[spec_xq<F_>()], px1[spec_xq<F_>()];
        float pq0[spec_tp<A_>()][4], pq1[spec_tp<A_>()][4];
        if constexpr (LOOP) {
            spec_fetch_x<DUAL, F_>(A.s[0].in0, A.s[0].in1, blockIdx.x * RC_ROWS, N, tid, px0, px1);
            spec_load_rows<A_>(A.s[1].res0, blockIdx.x * RC_ROWS + li, N, wid, lk, pq0);
            spec_load_rows<A_>(DUAL ? A.s[1].res1 : nullptr, blockIdx.x * RC_ROWS + li, N, wid, lk, pq1);
        }
        for (int tile = blockIdx.x, it = 0; LOOP ? tile < n_tiles : it < 1; tile += t_step, ++it) {
            const int row0 = tile * RC_ROWS, row = row0 + li;
            float q0[spec_tp<A_>()][4], q1[spec_tp<A_>()][4];
            if constexpr (LOOP) {
                spec_put_x<DUAL, F_, NP>(X0, X1, ldt, tid, px0, px1);
                spec_copy<A_>(q0, pq0); spec_copy<A_>(q1, pq1);
                const int nrow0 = (tile + t_step) * RC_ROWS;
                spec_fetch_x<DUAL, F_>(A.s[0].in0, A.s[0].in1, nrow0, N, tid, px0, px1);
                spec_load_rows<A_>(A.s[1].res0, nrow0 + li, N, wid, lk, pq0);
                spec_load_rows<A_>(DUAL ? A.s[1].res1 : nullptr, nrow0 + li, N, wid, lk, pq1);
            } else {
                spec_load_x<DUAL, F_, NP>(A.s[0].in0, A.s[0].in1, X0, X1, ldt, row0, N, tid);
                spec_load_rows<A_>(A.s[1].res0, row, N, wid, lk, q0);
                spec_load_rows<A_>(DUAL ? A.s[1].res1 : nullptr, row, N, wid, lk, q1);
            }
            f32x4 a0[spec_tp<A_>()], a1[spec_tp<A_>()];
            __syncthreads();
            // stage 0: t = ssp(U1 m + c1), su, td
            spec_mma<DUAL, F_, A_>(X0, X1, ldt, wid, li, lk, w0, a0, a1);
            __syncthreads();
            if constexpr (!LOOP) spec_load_bias<A_, MDG_CHAIN_NONE>(A.s[0], wid, lk, bA0, lA);
            spec_epilogue<DUAL, A_, 1, MDG_CHAIN_NONE, NP>(A.s[0], a0, a1, zA, zA, zA, zA, sgA, tdA, X0, X1, ldt, row, N, wid, li, lk, bA0, zA);
            __syncthreads();
            // stage 1: r' = U2 t + c2 + r
            spec_mma<DUAL, A_, A_>(X0, X1, ldt, wid, li, lk, w1, a0, a1);
            __syncthreads();
            {
                float s_[spec_tp<A_>()][4], t_[spec_tp<A_>()][4];
                if constexpr (!LOOP) spec_load_bias<A_, MDG_CHAIN_NONE>(A.s[1], wid, lk, bA1, lA);
                spec_epilogue<DUAL, A_, 0, MDG_CHAIN_NONE, NP>(A.s[1], a0, a1, zA, zA, q0, q1, s_, t_, X0, X1, ldt, row, N, wid, li, lk, bA1, zA);
            }
            __syncthreads();
            if constexpr (!LOOP) {
                spec_load_w<A_, H_, false>(A.s[2].W, wid, li, lk, w2);
                spec_load_w<H_, A_, true>(A.s[3].W, wid, li, lk, w3);
                spec_load_w<A_, A_, true>(A.s[4].W, wid, li, lk, w4);
                spec_load_w<A_, F_, true>(A.s[5].W, wid, li, lk, w5);
            }
            // stage 2: readout + head: sy, syd kept in global; out = (ydb, yb)
            f32x4 h0[spec_tp<H_>()], h1[spec_tp<H_>()];
            spec_mma<DUAL, A_, H_>(X0, X1, ldt, wid, li, lk, w2, h0, h1);
            __syncthreads();
            if constexpr (!LOOP) spec_load_bias<H_, MDG_CHAIN_HEAD>(A.s[2], wid, lk, bH, lH);
            spec_epilogue<DUAL, H_, 1, MDG_CHAIN_HEAD, NP>(A.s[2], h0, h1, zH, zH, zH, zH, dH, dH, X0, X1, ldt, row, N, wid, li, lk, bH, lH);
            __syncthreads();
            // stage 3: (rdb, rb) = (ydb, yb) L1
            spec_mma<DUAL, H_, A_>(X0, X1, ldt, wid, li, lk, w3, a0, a1);
            __syncthreads();
            {
                float s_[spec_tp<A_>()][4], t_[spec_tp<A_>()][4];
                spec_epilogue<DUAL, A_, 0, MDG_CHAIN_NONE, NP>(A.s[3], a0, a1, zA, zA, zA, zA, s_, t_, X0, X1, ldt, row, N, wid, li, lk, zA, zA);
            }
            __syncthreads();
            // stage 4: (tdb, tb) = (rdb, rb) U2, then the reverse of the (ssp, tangent) pair with su / td of stage 0 (registers)
            spec_mma<DUAL, A_, A_>(X0, X1, ldt, wid, li, lk, w4, a0, a1);
            __syncthreads();
            {
                float s_[spec_tp<A_>()][4], t_[spec_tp<A_>()][4];
                spec_epilogue<DUAL, A_, 0, DUAL ? MDG_CHAIN_SSP_BWD : MDG_CHAIN_MUL, NP>(A.s[4], a0, a1, sgA, tdA, zA, zA, s_, t_, X0, X1, ldt, row,
                                                                                    N, wid, li, lk, zA, zA);
            }
            __syncthreads();
            // stage 5: (mdb, mb) = (udb, ub) U1
            f32x4 f0[spec_tp<F_>()], f1[spec_tp<F_>()];
            spec_mma<DUAL, A_, F_>(X0, X1, ldt, wid, li, lk, w5, f0, f1);
            __syncthreads();
            spec_epilogue<DUAL, F_, 0, MDG_CHAIN_NONE, NP>(A.s[5], f0, f1, zF, zF, zF, zF, dF, dF, X0, X1, ldt, row, N, wid, li, lk, zF, zF);
            if constexpr (LOOP) __syncthreads();
        }
    } else {
        // ---- REV: (rdb', rb') = (hdb, hb) Wn + (rdb, rb); (udb, ub) = ssp'((rdb', rb') U2); (mdb, mb) = (udb, ub) U1
        WFrag<NP, F_, A_> w0; WFrag<NP, A_, A_> w1; WFrag<NP, A_, F_> w2;
        spec_load_w<F_, A_, true>(A.s[0].W, wid, li, lk, w0);
        spec_load_w<A_, A_, true>(A.s[1].W, wid, li, lk, w1);
        spec_load_w<A_, F_, true>(A.s[2].W, wid, li, lk, w2);
        float4 px0[spec_xq<F_>()], px1[spec_xq<F_>()];
        float pq0[spec_tp<A_>()][4], pq1[spec_tp<A_>()][4], psg[spec_tp<A_>()][4], ptd[spec_tp<A_>()][4];
        if constexpr (LOOP) {
            const int frow = blockIdx.x * RC_ROWS + li;
            spec_fetch_x<DUAL, F_>(A.s[0].in0, A.s[0].in1, blockIdx.x * RC_ROWS, N, tid, px0, px1);
            spec_load_rows<A_>(A.s[0].res0, frow, N, wid, lk, pq0);
            spec_load_rows<A_>(DUAL ? A.s[0].res1 : nullptr, frow, N, wid, lk, pq1);
            spec_load_rows<A_>(A.s[1].aux0, frow, N, wid, lk, psg);
            spec_load_rows<A_>(DUAL ? A.s[1].aux1 : nullptr, frow, N, wid, lk, ptd);
        }
        for (int tile = blockIdx.x, it = 0; LOOP ? tile < n_tiles : it < 1; tile += t_step, ++it) {
            const int row0 = tile * RC_ROWS, row = row0 + li;
            float q0[spec_tp<A_>()][4], q1[spec_tp<A_>()][4];
            if constexpr (LOOP) {
                spec_put_x<DUAL, F_, NP>(X0, X1, ldt, tid, px0, px1);
                spec_copy<A_>(q0, pq0); spec_copy<A_>(q1, pq1); spec_copy<A_>(sgA, psg); spec_copy<A_>(tdA, ptd);
                const int nrow0 = (tile + t_step) * RC_ROWS, nrow = nrow0 + li;
                spec_fetch_x<DUAL, F_>(A.s[0].in0, A.s[0].in1, nrow0, N, tid, px0, px1);
                spec_load_rows<A_>(A.s[0].res0, nrow, N, wid, lk, pq0);
                spec_load_rows<A_>(DUAL ? A.s[0].res1 : nullptr, nrow, N, wid, lk, pq1);
                spec_load_rows<A_>(A.s[1].aux0, nrow, N, wid, lk, psg);
                spec_load_rows<A_>(DUAL ? A.s[1].aux1 : nullptr, nrow, N, wid, lk, ptd);
            } else {
                spec_load_x<DUAL, F_, NP>(A.s[0].in0, A.s[0].in1, X0, X1, ldt, row0, N, tid);
                spec_load_rows<A_>(A.s[0].res0, row, N, wid, lk, q0);
                spec_load_rows<A_>(DUAL ? A.s[0].res1 : nullptr, row, N, wid, lk, q1);
                spec_load_rows<A_>(A.s[1].aux0, row, N, wid, lk, sgA);
                spec_load_rows<A_>(DUAL ? A.s[1].aux1 : nullptr, row, N, wid, lk, tdA);
            }
            f32x4 a0[spec_tp<A_>()], a1[spec_tp<A_>()];
            float s_[spec_tp<A_>()][4], t_[spec_tp<A_>()][4];
            __syncthreads();
            spec_mma<DUAL, F_, A_>(X0, X1, ldt, wid, li, lk, w0, a0, a1);
            __syncthreads();
            spec_epilogue<DUAL, A_, 0, MDG_CHAIN_NONE, NP>(A.s[0], a0, a1, zA, zA, q0, q1, s_, t_, X0, X1, ldt, row, N, wid, li, lk, zA, zA);
            __syncthreads();
            spec_mma<DUAL, A_, A_>(X0, X1, ldt, wid, li, lk, w1, a0, a1);
            __syncthreads();
            spec_epilogue<DUAL, A_, 0, DUAL ? MDG_CHAIN_SSP_BWD : MDG_CHAIN_MUL, NP>(A.s[1], a0, a1, sgA, tdA, zA, zA, s_, t_, X0, X1, ldt, row, N, wid,
                                                                                li, lk, zA, zA);
            __syncthreads();
            f32x4 f0[spec_tp<F_>()], f1[spec_tp<F_>()];
            spec_mma<DUAL, A_, F_>(X0, X1, ldt, wid, li, lk, w2, f0, f1);
            __syncthreads();
            spec_epilogue<DUAL, F_, 0, MDG_CHAIN_NONE, NP>(A.s[2], f0, f1, zF, zF, zF, zF, dF, dF, X0, X1, ldt, row, N, wid, li, lk, zF, zF);
            if constexpr (LOOP) __syncthreads();
        }
    }
}

// does the descriptor list have the shape of one of the compiled chains (widths A, F)?  -1: no
int spec_kind(const MdgChainStage* s, int n, int dual, int A_, int F_) {
    const int H_ = A_ / 2;
    auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
    for (int i = 0; i < n; ++i) {
        const MdgChainStage& t = s[i];
        if (!al(t.W) || !al(t.bias) || !al(t.in0) || !al(t.in1) || !al(t.res0) || !al(t.res1) || !al(t.aux0) || !al(t.aux1) ||
            !al(t.out0) || !al(t.out1) || !al(t.sig) || !al(t.pre0) || !al(t.pre1))
            return -1;
        if (i > 0 && t.in0) return -1;
    }
    auto plain = [](const MdgChainStage& t) { return t.mode == MDG_CHAIN_NONE && !t.aux0 && !t.aux1; };
    auto st = [](const MdgChainStage& t, int K, int M, int trans, int act) { return t.K == K && t.M == M && t.trans == trans && t.act == act; };
    const int bmode = dual ? MDG_CHAIN_SSP_BWD : MDG_CHAIN_MUL;
    if (n == 3 && st(s[0], F_, A_, 0, 1) && plain(s[0]) && !s[0].res0 && !s[0].res1 && st(s[1], A_, A_, 0, 0) && plain(s[1]) &&
        st(s[2], A_, F_, 0, 0) && plain(s[2]) && !s[2].res0 && !s[2].res1)
        return SPEC_FWD;
    if (n == 6 && st(s[0], F_, A_, 0, 1) && plain(s[0]) && !s[0].res0 && !s[0].res1 && st(s[1], A_, A_, 0, 0) && plain(s[1]) &&
        st(s[2], A_, H_, 0, 1) && s[2].mode == MDG_CHAIN_HEAD && !s[2].res0 && !s[2].res1 && st(s[3], H_, A_, 1, 0) && plain(s[3]) &&
        !s[3].bias && !s[3].res0 && !s[3].res1 && st(s[4], A_, A_, 1, 0) && s[4].mode == bmode && s[4].aux0 == s[0].sig && s[0].sig &&
        (!dual || (s[4].aux1 == s[0].out1 && s[0].out1)) && !s[4].bias && !s[4].res0 && !s[4].res1 && st(s[5], A_, F_, 1, 0) &&
        plain(s[5]) && !s[5].bias && !s[5].res0 && !s[5].res1)
        return SPEC_TURN;
    if (n == 3 && st(s[0], F_, A_, 1, 0) && plain(s[0]) && !s[0].bias && st(s[1], A_, A_, 1, 0) && s[1].mode == bmode && s[1].aux0 &&
        (!dual || s[1].aux1) && !s[1].bias && !s[1].res0 && !s[1].res1 && st(s[2], A_, F_, 1, 0) && plain(s[2]) && !s[2].bias &&
        !s[2].res0 && !s[2].res1)
        return SPEC_REV;
    return -1;
}

// workgroups of the LOOP instantiation: two per CU -- what the hoisted weight fragments leave room for (148 - 219 registers) --
// trimmed so that every workgroup walks the same number of tiles where that is possible; 0: one round of workgroups, no loop
// (fewer than two tiles per looping workgroup, or MDG_CHAIN_LOOP_WGS=0; MDG_CHAIN_LOOP_WGS=n: n workgroups per CU).
// Measured at 32 768 rows (tools/kbench_chain.py): dual forward chain 42.3 -> 38.4 us, dual turn chain 60.9 -> 53.4 us; three
// per CU leaves a third of the CUs with one workgroup more than the rest (40.4 / 61.2 us).
int loop_grid(int n_tiles) {
    static int per_cu = -1, cus = 0;
    if (per_cu < 0) {
        const char* e = getenv("MDG_CHAIN_LOOP_WGS");
        int dev = 0;
        hipDeviceProp_t p;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ? p.multiProcessorCount : 256;
        per_cu = e ? atoi(e) : 2;
    }
    if (per_cu <= 0 || n_tiles < 2 * per_cu * cus) return 0;
    const int rounds = (n_tiles + per_cu * cus - 1) / (per_cu * cus);
    return (n_tiles + rounds - 1) / rounds;
}

template <int A_, int F_, bool DUAL, int KIND, int NP>
void spec_launch_kind(const ChainArgs& a, int n_tiles, hipStream_t st) {
    int lg = 0;
    // (LOOP: n_atom_basis = 64, whose weight fragments fit the registers of two workgroups per CU -- with three planes, X6, the
    //  turn chain's do not: it runs one round of workgroups)
    constexpr bool can_loop = A_ == 64 && !(NP == 3 && KIND == SPEC_TURN);
    if constexpr (can_loop) lg = loop_grid(n_tiles);
    if constexpr (can_loop) {
        if (lg) { hipLaunchKernelGGL((chain_spec_kernel<A_, F_, DUAL, KIND, true, NP>), dim3(lg), dim3(256), 0, st, a); return; }
    }
    hipLaunchKernelGGL((chain_spec_kernel<A_, F_, DUAL, KIND, false, NP>), dim3(n_tiles), dim3(256), 0, st, a);
}

// np: planes per operand the caller asked for (0 / 2: MDG_CHAIN_X3 / 3: MDG_CHAIN_X6)
template <int A_, int F_>
bool spec_launch(const ChainArgs& a, const MdgChainStage* s, int n, int n_rows, int dual, int np, hipStream_t st) {
    const int kind = spec_kind(s, n, dual, A_, F_);
    if (kind < 0) return false;
    const int n_tiles = (n_rows + RC_ROWS - 1) / RC_ROWS;
#define MDG_SPEC_NP(K_, NP_) do { if (dual) spec_launch_kind<A_, F_, true, K_, NP_>(a, n_tiles, st); else spec_launch_kind<A_, F_, false, K_, NP_>(a, n_tiles, st); } while (0)
#define MDG_SPEC(K_)                                                                                                   \
    do {                                                                                                               \
        /* X6 is as accurate as the f32 instruction, so it is taken where it is faster (tools/kbench_chain.py --flag 4):  \
           few tiles (latency-bound: 192 rows, A = 128: 18.3 / 26.1 -> 14.5 / 21.2 us), the looping A = 64 chains (39.7 ->  \
           33.0 us at 32 768 rows) and the dual A = 128 turn chain (80 -> 67 us at 12 288 rows); elsewhere the extra    \
           registers cost more than the matrix time saved */                                                           \
        if (np == 3 && (n_tiles <= 512 || (A_ == 64 && K_ != SPEC_TURN) || (A_ == 128 && dual && K_ == SPEC_TURN))) {     \
            MDG_SPEC_NP(K_, 3);                                                                                        \
            break;                                                                                                     \
        }                                                                                                              \
        if constexpr (A_ == 64) {                          /* X3 is compiled for the n_atom_basis = 64 chains */        \
            if (np == 2) { MDG_SPEC_NP(K_, 2); break; }                                                                \
        }                                                                                                              \
        MDG_SPEC_NP(K_, 0);                                                                                            \
    } while (0)
    if (kind == SPEC_FWD) MDG_SPEC(SPEC_FWD); else if (kind == SPEC_TURN) MDG_SPEC(SPEC_TURN); else MDG_SPEC(SPEC_REV);
#undef MDG_SPEC
#undef MDG_SPEC_NP
    return true;
}

}  // namespace

extern "C" int mdg_row_chain(const MdgChainStage* stages, int n_stages, int n_rows, int flags, void* stream) {
    MDG_CHECK_ARG(flags >= 0 && flags <= (MDG_CHAIN_DUAL | MDG_CHAIN_X3 | MDG_CHAIN_X6) && (flags & (MDG_CHAIN_X3 | MDG_CHAIN_X6)) != (MDG_CHAIN_X3 | MDG_CHAIN_X6),
                  "row_chain: flags: MDG_CHAIN_DUAL | one of MDG_CHAIN_X3, MDG_CHAIN_X6");
    const int dual = flags & MDG_CHAIN_DUAL;
    // (X3: honoured by the compiled n_atom_basis = 64 chains, X6: by every compiled chain; f32 matrix instruction elsewhere)
    const int x3 = (flags & MDG_CHAIN_X6) ? 3 : ((flags & MDG_CHAIN_X3) ? 2 : 0);
    MDG_CHECK_ARG(stages && n_stages >= 1 && n_stages <= MDG_CHAIN_MAX_STAGES, "row_chain: 1..%d stages", MDG_CHAIN_MAX_STAGES);
    MDG_CHECK_ARG(n_rows >= 0 && (int64_t)n_rows * MDG_CHAIN_MAX_WIDTH < ((int64_t)1 << 31), "row_chain: bad row count");
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
        MDG_CHECK_ARG((!s.out0_h && !s.out1_h) || (s.out0_h && s.M % 4 == 0 && (!dual || s.out1_h) && (((uintptr_t)s.out0_h | (uintptr_t)s.out1_h) & 7) == 0),
                      "row_chain: stage %d: bf16 mirrors come for both outputs, 8-byte aligned, widths in multiples of 4", i);
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
    if (!getenv("MDG_CHAIN_WALKER")) {                                  // (MDG_CHAIN_WALKER=1: always the descriptor walker)
        if (spec_launch<64, 128>(a, stages, n_stages, n_rows, dual, x3, st) || spec_launch<128, 128>(a, stages, n_stages, n_rows, dual, x3, st) ||
            spec_launch<64, 64>(a, stages, n_stages, n_rows, dual, x3, st) || spec_launch<128, 64>(a, stages, n_stages, n_rows, dual, x3, st)) {
            MDG_CHECK_LAUNCH("chain_spec_kernel");
            return MDG_OK;
        }
    }
    const bool lat = n_rows <= 8192;
#define MDG_RC(D_, T_) do { if (lat) hipLaunchKernelGGL((row_chain_kernel<D_, T_, true>), grid, block, lds, st, a); \
                            else hipLaunchKernelGGL((row_chain_kernel<D_, T_, false>), grid, block, lds, st, a); } while (0)
    if (dual) { if (tiles <= 2) MDG_RC(true, 2); else MDG_RC(true, 8); }
    else { if (tiles <= 2) MDG_RC(false, 2); else MDG_RC(false, 8); }
#undef MDG_RC
    MDG_CHECK_LAUNCH("row_chain_kernel");
    return MDG_OK;
}
