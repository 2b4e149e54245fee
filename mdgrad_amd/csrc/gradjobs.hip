// Batched parameter-gradient reductions of one SchNet adjoint evaluation (mdgrad_amd/nn/analytic.py).
//
// The reverse sweep of U_dot (what double autograd derives at torchmd/sovlers.py:229-233 for the parameters of
// nff/nn/modules.py:514-575 and nff/nn/models/schnet.py:113-171) leaves ~25 small reductions over the atoms: the
// weight gradients  x^T g (+ xd^T gd)  of every node-level Dense layer (tall-skinny [N,M]^T [N,K] products), the bias
// gradients (column sums, some of an elementwise product), the rows of the embedding table, and the filter-network
// gradients that cfconv_bwd already reduced.  Issued one by one they were ~50 launches per evaluation (split-K product +
// reduce per weight, torch reductions / cat / neg / scale / add for the rest); here ALL of them are TWO launches:
//
//   grad_partial_kernel   every (job, 64 x 64 unit, K-slab) on its own workgroup: products on v_mfma_f32_16x16x4_f32 with
//                         16-byte operand loads (a lane's float4 feeds four 16x16 tiles: the k index of an MFMA operand
//                         is the row of the tall matrix, the 16 "column" lanes take 4 consecutive floats each), the four
//                         waves interleave the slab's rows and combine in LDS; column sums by row-strided float4 reads +
//                         an ordered LDS combine; partial results to a workspace;
//   grad_reduce_kernel    fixed-order sum over the slabs and  flat[dst] (+)= alpha * (t[i] - t[i-1]) * value: the
//                         interval weight of sovlers.py:160 is read on the device, the destination is the caller's flat
//                         gradient buffer in nn.Module.parameters() order (tinydiffeq.py:106-108).
//
// No atomics; two launches with the same inputs give the same bits.
#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct JobTable {
    MdgGradJob j[MDG_GRAD_JOBS_MAX];
    long long ws_off[MDG_GRAD_JOBS_MAX];      // first float of the job's partial results in the workspace
    long long slab[MDG_GRAD_JOBS_MAX];        // rows per K-slab
    int first_block[MDG_GRAD_JOBS_MAX + 1];   // partial kernel: blocks [first_block[k], first_block[k+1]) belong to job k
    int splits[MDG_GRAD_JOBS_MAX];
    int first_out[MDG_GRAD_JOBS_MAX + 1];     // reduce kernel: output elements (padded to 64 per job), prefix sums
    int n_jobs;
};

__device__ __forceinline__ f32x4 load4(const float* __restrict__ p, long long row, int width, int col0, bool row_ok) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (!row_ok) return v;
    const float* q = p + row * (long long)width + col0;
    if ((width & 3) == 0 && col0 + 3 < width) {
        v = *reinterpret_cast<const f32x4*>(q);
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (col0 + c < width) v[c] = q[c];
    }
    return v;
}

// One wave: its share of C[64 x 64] = A[:, m0:m0+64]^T B[:, n0:n0+64] over the rows k0 + 4 (4 i + w) + {0..3}, i = 0, 1, ...
// (the four waves of a workgroup interleave 4-row steps of the workgroup's K-slab).  Tile (c, c2): rows m = m0 + 4 i + c,
// columns n = n0 + 4 i2 + c2 (i, i2 = the MFMA's 16 row / column lanes), so a lane's float4 loads feed four tiles each.
// Two steps of loads are in flight while the 16 MFMAs of the current one run.
__device__ __forceinline__ void atb_wave(const MdgGradJob& J, int m0, int n0, long long k0, long long k1, int w, f32x4 (&acc)[4][4]) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int M = J.m, N = J.n;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) acc[c][c2] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int pass = 0; pass < (J.A2 ? 2 : 1); ++pass) {
        const float* __restrict__ Ap = pass ? J.A2 : J.A;
        const float* __restrict__ Bp = pass ? J.B2 : J.B;
        const long long kw = k0 + 4 * w;                          // the wave's step 0 (uniform); this lane's row: + lk
        f32x4 a0 = load4(Ap, kw + lk, M, m0 + 4 * li, kw + lk < k1), b0 = load4(Bp, kw + lk, N, n0 + 4 * li, kw + lk < k1);
        f32x4 a1 = load4(Ap, kw + 16 + lk, M, m0 + 4 * li, kw + 16 + lk < k1);
        f32x4 b1 = load4(Bp, kw + 16 + lk, N, n0 + 4 * li, kw + 16 + lk < k1);
        for (long long kb = kw; kb < k1; kb += 16) {              // (uniform trip count over the wave: rows are masked)
            const long long r2 = kb + 32 + lk;
            const f32x4 a2 = load4(Ap, r2, M, m0 + 4 * li, r2 < k1), b2 = load4(Bp, r2, N, n0 + 4 * li, r2 < k1);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int c2 = 0; c2 < 4; ++c2)
                    acc[c][c2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[c], b0[c2], acc[c][c2], 0, 0, 0);
            a0 = a1; b0 = b1; a1 = a2; b1 = b2;
        }
    }
}

__global__ __launch_bounds__(256) void grad_partial_kernel(const JobTable T, float* __restrict__ ws) {
    __shared__ __attribute__((aligned(16))) float atb_red[3 * 16 * 64 * 4];        // 48 KB: three waves' accumulators
    f32x4* red = reinterpret_cast<f32x4*>(atb_red);                                  // (column sums: 256 x float4)
    int k = 0;
    while (k + 1 < T.n_jobs && (int)blockIdx.x >= T.first_block[k + 1]) ++k;
    const MdgGradJob& J = T.j[k];
    const int local = (int)blockIdx.x - T.first_block[k];
    const int wid = threadIdx.x >> 6;
    if (J.kind == MDG_GRAD_ATB) {
        // one workgroup = one (64 x 64 unit, K-slab): the four waves take interleaved 4-row steps of the slab, their
        // accumulators are combined through LDS in wave order, wave 0 writes the partial block
        const int units_n = (J.n + 63) / 64, units = ((J.m + 63) / 64) * units_n;
        const int unit = local % units, split = local / units;
        const long long k0 = (long long)split * T.slab[k], k1 = min(J.rows, k0 + T.slab[k]);
        const int m0 = (unit / units_n) * 64, n0 = (unit % units_n) * 64;
        f32x4 acc[4][4];
        atb_wave(J, m0, n0, k0, k1, wid, acc);
        const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
        f32x4* buf = reinterpret_cast<f32x4*>(atb_red);           // [3 waves][16 tiles][64 lanes]
        if (wid > 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int c2 = 0; c2 < 4; ++c2) buf[((wid - 1) * 16 + c * 4 + c2) * 64 + lane] = acc[c][c2];
        }
        __syncthreads();
        if (wid != 0) return;
        float* out = ws + T.ws_off[k] + (size_t)split * J.m * J.n;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2) {
                f32x4 v = acc[c][c2];
#pragma unroll
                for (int q = 0; q < 3; ++q) v += buf[(q * 16 + c * 4 + c2) * 64 + lane];
                const int n = n0 + 4 * li + c2;
                if (n >= J.n) continue;
                // accumulator layout: column lane = lane & 15, row = (lane >> 4) * 4 + r
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + 4 * (lk * 4 + r) + c;
                    if (m < J.m) out[(size_t)m * J.n + n] = v[r];
                }
            }
        return;
    }
    // column sums of A (.* B) (+ A2 .* B2) over the block's row slab: a thread owns 4 consecutive columns, the block's
    // 256 / (cols / 4) row lanes walk the slab with that stride; ordered combine through LDS
    const int groups = (J.m + 3) / 4;                              // float4 column groups (<= 256: m <= 1024)
    const int cg_per_blk = min(groups, 256), rl_n = 256 / cg_per_blk;
    const int col_blocks = (groups + cg_per_blk - 1) / cg_per_blk;
    const int split = local / col_blocks, cb = local % col_blocks;
    const int cg = cb * cg_per_blk + (int)(threadIdx.x % cg_per_blk), rl = (int)(threadIdx.x / cg_per_blk);
    const long long k0 = (long long)split * T.slab[k], k1 = min(J.rows, k0 + T.slab[k]);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (cg < groups && rl < rl_n) {
        for (int pass = 0; pass < (J.A2 ? 2 : 1); ++pass) {
            const float* __restrict__ Ap = pass ? J.A2 : J.A;
            const float* __restrict__ Bp = pass ? J.B2 : J.B;
            // four rows in flight per thread (the loads of one row are a dependent round trip otherwise)
            long long r = k0 + rl;
            f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1, s3 = s1;
            for (; r + 3 * rl_n < k1; r += 4 * rl_n) {
                f32x4 a0 = load4(Ap, r, J.m, 4 * cg, true), a1 = load4(Ap, r + rl_n, J.m, 4 * cg, true);
                f32x4 a2 = load4(Ap, r + 2 * rl_n, J.m, 4 * cg, true), a3 = load4(Ap, r + 3 * rl_n, J.m, 4 * cg, true);
                if (Bp) {
                    a0 *= load4(Bp, r, J.m, 4 * cg, true); a1 *= load4(Bp, r + rl_n, J.m, 4 * cg, true);
                    a2 *= load4(Bp, r + 2 * rl_n, J.m, 4 * cg, true); a3 *= load4(Bp, r + 3 * rl_n, J.m, 4 * cg, true);
                }
                s += a0; s1 += a1; s2 += a2; s3 += a3;
            }
            for (; r < k1; r += rl_n) {
                f32x4 a = load4(Ap, r, J.m, 4 * cg, true);
                if (Bp) a *= load4(Bp, r, J.m, 4 * cg, true);
                s += a;
            }
            s += (s1 + s2) + s3;
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0 && cg < groups) {
        for (int q = 1; q < rl_n; ++q) s += red[q * cg_per_blk + (int)(threadIdx.x % cg_per_blk)];
        float* out = ws + T.ws_off[k] + (size_t)split * J.m;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (4 * cg + c < J.m) out[4 * cg + c] = s[c];
    }
}

__global__ __launch_bounds__(256) void grad_reduce_kernel(const JobTable T, const float* __restrict__ ws, float* __restrict__ flat,
                                                          float alpha, const float* __restrict__ tgrid,
                                                          const long long* __restrict__ idx, int accumulate) {
    // a block covers 64 consecutive output elements of ONE job (every job's outputs are padded to a multiple of 64); its
    // four waves take the slabs q = wave, wave + 4, ... -- rows of the workspace, read coalesced and four at a time -- and
    // add up in wave order
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int g0 = (int)blockIdx.x * 64, g = g0 + lane;
    int k = 0;
    while (k + 1 < T.n_jobs && g0 >= T.first_out[k + 1]) ++k;
    const MdgGradJob& J = T.j[k];
    const int e = g - T.first_out[k];
    const int MN = J.kind == MDG_GRAD_ATB ? J.m * J.n : J.m;
    const bool live = e < MN;
    float s = 0.f;
    if (live) {
        if (J.kind == MDG_GRAD_AXPY) {
            if (sub == 0) s = J.A[e] + (J.A2 ? J.A2[e] : 0.f);
        } else {
            const float* p = ws + T.ws_off[k] + e;
            const int ns = T.splits[k];
            float s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int q = sub;
            for (; q + 12 < ns; q += 16) {
                s += p[(size_t)q * MN]; s1 += p[(size_t)(q + 4) * MN]; s2 += p[(size_t)(q + 8) * MN]; s3 += p[(size_t)(q + 12) * MN];
            }
            for (; q < ns; q += 4) s += p[(size_t)q * MN];
            s = (s + s1) + (s2 + s3);
        }
    }
    red[sub][lane] = s;
    __syncthreads();
    if (!live || sub != 0) return;
    s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    float f = alpha;
    if (tgrid) { const long long i = idx[0]; f *= tgrid[i] - tgrid[i - 1]; }
    long long dst = J.out_off + e;
    if (J.kind == MDG_GRAD_ATB && J.row_map) dst = J.out_off + (long long)J.row_map[e / J.n] * J.n + (e % J.n);
    flat[dst] = accumulate ? fmaf(f, s, flat[dst]) : f * s;
}

int plan(const MdgGradJob* jobs, int n_jobs, JobTable& T, long long& ws_floats) {
    if (n_jobs < 0 || n_jobs > MDG_GRAD_JOBS_MAX) { mdg_set_error("grad_jobs: at most %d jobs per call", MDG_GRAD_JOBS_MAX); return MDG_EINVAL; }
    T.n_jobs = n_jobs;
    int n_atb = 0;
    for (int k = 0; k < n_jobs; ++k) n_atb += jobs[k].kind == MDG_GRAD_ATB;
    const int atb_budget = 4096 / (n_atb > 0 ? n_atb : 1);
    // Small calls (one system of a few thousand atoms): every workgroup of the launch holds the 48 KB of the product's LDS
    // combine, i.e. three are resident per CU -- 768 on the device.  When the column-sum workgroups leave room, the products
    // share what is left of that ONE resident round (same number of k-slabs each) instead of spilling into a second one
    // (4 096 rows: 848 workgroups -> 764, 22.7 -> see tools/kbench_gradjobs.py).
    long long cs_blocks = 0, atb_units = 0;
    for (int k = 0; k < n_jobs; ++k) {
        const MdgGradJob& J = jobs[k];
        if (J.kind == MDG_GRAD_COLSUM && J.m > 0) {
            const int groups = (J.m + 3) / 4, cgb = groups < 256 ? groups : 256;
            long long want = (J.rows + 127) / 128;
            want = want < 1 ? 1 : (want > 256 ? 256 : want);
            cs_blocks += want * ((groups + cgb - 1) / cgb);
        } else if (J.kind == MDG_GRAD_ATB && J.m > 0 && J.n > 0) {
            atb_units += (long long)((J.m + 63) / 64) * ((J.n + 63) / 64);
        }
    }
    const long long one_round = 768;
    const long long splits_one_round = atb_units > 0 && cs_blocks < one_round ? (one_round - cs_blocks) / atb_units : 0;
    long long ws = 0;
    int blocks = 0, outs = 0;
    for (int k = 0; k < n_jobs; ++k) {
        const MdgGradJob& J = jobs[k];
        T.j[k] = J;
        T.first_block[k] = blocks;
        T.first_out[k] = outs;
        T.ws_off[k] = ws;
        const bool pair_ok = J.kind == MDG_GRAD_AXPY ? true
                           : J.kind == MDG_GRAD_ATB ? (J.A2 == nullptr) == (J.B2 == nullptr)
                                                    : (J.A2 == nullptr ? J.B2 == nullptr : (J.B2 == nullptr) == (J.B == nullptr));
        if (J.m <= 0 || J.rows < 0 || !J.A || !pair_ok) {
            mdg_set_error("grad_jobs: job %d has bad operands", k);
            return MDG_EINVAL;
        }
        if (J.kind == MDG_GRAD_ATB) {
            if (J.n <= 0 || !J.B) { mdg_set_error("grad_jobs: job %d: a product needs B and n > 0", k); return MDG_EINVAL; }
            const int units = ((J.m + 63) / 64) * ((J.n + 63) / 64);
            long long want = (atb_budget + units - 1) / units;            // workgroups of this job: its share of ~2 rounds of the chip
            // at least 64 rows (4 steps per wave) per slab: the product is bound by the latency of its row loads (two steps in
            // flight per wave), so short slabs on many workgroups beat long ones -- 4 096 rows: 50 us at 256 rows per slab
            const long long maxs = (J.rows + 63) / 64;
            if (want > maxs) want = maxs;
            if (splits_one_round >= 8 && want > splits_one_round) want = splits_one_round;
            const long long cap = ((long long)1 << 20) / ((long long)J.m * J.n);      // <= 4 MB of partial blocks per job
            if (want > cap && cap >= 16) want = cap;
            if (want < 1) want = 1;
            if (want > 512) want = 512;
            long long slab = (J.rows + want - 1) / want;
            slab = (slab + 15) / 16 * 16;
            if (slab < 16) slab = 16;
            const int splits = (int)((J.rows + slab - 1) / slab > 0 ? (J.rows + slab - 1) / slab : 1);
            T.splits[k] = splits;
            T.slab[k] = slab;
            blocks += units * splits;
            ws += (long long)splits * J.m * J.n;
            outs += (J.m * J.n + 63) / 64 * 64;
        } else if (J.kind == MDG_GRAD_COLSUM) {
            if (J.m > 1024) { mdg_set_error("grad_jobs: job %d: column sums take at most 1024 columns", k); return MDG_EINVAL; }
            const int groups = (J.m + 3) / 4, cg_per_blk = groups < 256 ? groups : 256;
            const int col_blocks = (groups + cg_per_blk - 1) / cg_per_blk;
            long long want = (J.rows + 127) / 128;                      // ~128 rows per block
            if (want < 1) want = 1;
            if (want > 256) want = 256;
            const long long slab = (J.rows + want - 1) / want > 0 ? (J.rows + want - 1) / want : 1;
            const int splits = (int)((J.rows + slab - 1) / slab > 0 ? (J.rows + slab - 1) / slab : 1);
            T.splits[k] = splits;
            T.slab[k] = slab;
            blocks += splits * col_blocks;
            ws += (long long)splits * J.m;
            outs += (J.m + 63) / 64 * 64;
        } else if (J.kind == MDG_GRAD_AXPY) {
            T.splits[k] = 0;
            T.slab[k] = 0;
            outs += (J.m + 63) / 64 * 64;
        } else {
            mdg_set_error("grad_jobs: job %d: unknown kind %d", k, J.kind);
            return MDG_EINVAL;
        }
    }
    T.first_block[n_jobs] = blocks;
    T.first_out[n_jobs] = outs;
    ws_floats = ws;
    return MDG_OK;
}

}  // namespace

extern "C" int64_t mdg_grad_jobs_workspace(const MdgGradJob* jobs, int n_jobs) {
    JobTable T;
    long long ws = 0;
    if (plan(jobs, n_jobs, T, ws) != MDG_OK) return -1;
    return ws > 0 ? ws : 1;
}

extern "C" int mdg_grad_jobs(const MdgGradJob* jobs, int n_jobs, float* flat, float alpha, const float* t, const int64_t* idx,
                             int accumulate, float* workspace, void* stream) {
    MDG_CHECK_ARG(flat && (n_jobs == 0 || jobs), "grad_jobs: null buffer");
    MDG_CHECK_ARG((t == nullptr) == (idx == nullptr), "grad_jobs: the time grid and the frame index go together");
    if (n_jobs == 0) return MDG_OK;
    JobTable T;
    long long ws = 0;
    const int rc = plan(jobs, n_jobs, T, ws);
    if (rc != MDG_OK) return rc;
    MDG_CHECK_ARG(ws == 0 || workspace, "grad_jobs: null workspace");
    hipStream_t st = (hipStream_t)stream;
    if (T.first_block[n_jobs] > 0)
        hipLaunchKernelGGL(grad_partial_kernel, dim3(T.first_block[n_jobs]), dim3(256), 0, st, T, workspace);
    const int outs = T.first_out[n_jobs];
    hipLaunchKernelGGL(grad_reduce_kernel, dim3(outs / 64), dim3(256), 0, st, T, (const float*)workspace, flat, alpha, t,
                       (const long long*)idx, accumulate);
    MDG_CHECK_LAUNCH("grad_jobs kernels");
    return MDG_OK;
}
