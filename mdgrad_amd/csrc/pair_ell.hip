// K2-K4: pair energy / gradient / Hessian-vector product over the per-atom (ELL) list.
// Replaces compute_dis + pair form + .sum() and both autograd passes through them
// (torchmd/topology.py:5-12, torchmd/interface.py:298-299, torchmd/md.py:227-228,
//  torchmd/sovlers.py:229-233).
//
// LPA lanes cooperate on one atom: each lane walks a strided slice of the atom's (sorted)
// neighbour row, the per-atom sums are combined with wave shuffles ("wavefront-level
// segmented reduction"), scalar sums (energy, parameter gradients) go through a fixed-order
// block reduction into a per-block partial and a second tiny kernel adds the partials in
// block order.  No float atomics anywhere => bitwise reproducible.
#include "common.hpp"

namespace {

constexpr int NSCAL = 1 + 2 * MDG_MAX_THETA;   // energy, gtheta[3], gtheta_w[3]

struct EllArgs {
    const float* pos; int N; MdgCell cell;
    const int32_t* col; const int32_t* shift; const int32_t* cnt; int max_nbr;
    MdgPairTerm term; const float* theta; const float* w;
    float* grad; float* hw; float* partial;
    float oscale; int oacc;      // grad / hw outputs: out = (oacc ? out : 0) + oscale * value  (force sums of a Stack)
    int recheck;                 // the list was searched with a skin: re-apply the builders' exact cutoff test per pair
};

template <int LPA, int LEVEL>
__global__ void pair_ell_kernel(const EllArgs A) {
    __shared__ float red[16 * NSCAL];
    const int apb = blockDim.x / LPA;
    const int i = blockIdx.x * apb + threadIdx.x / LPA, sub = threadIdx.x % LPA;
    float vals[NSCAL];
#pragma unroll
    for (int k = 0; k < NSCAL; ++k) vals[k] = 0.f;
    if (i < A.N) {
        const TermConst tc = term_prepare(A.term, A.theta);
        const float xi = A.pos[3 * i], yi = A.pos[3 * i + 1], zi = A.pos[3 * i + 2];
        float wxi = 0.f, wyi = 0.f, wzi = 0.f;
        if (LEVEL >= 2) { wxi = A.w[3 * i]; wyi = A.w[3 * i + 1]; wzi = A.w[3 * i + 2]; }
        float gx = 0.f, gy = 0.f, gz = 0.f, hx = 0.f, hy = 0.f, hz = 0.f;
        const int n = A.cnt[i];
        const size_t row = (size_t)i * A.max_nbr;
        for (int k = sub; k < n; k += LPA) {
            const int j = A.col[row + k];
            float dx = xi - A.pos[3 * j], dy = yi - A.pos[3 * j + 1], dz = zi - A.pos[3 * j + 2];
            if (A.recheck) {
                // the test of the list builders at the current positions (D = x_j - x_i, reference minimum image,
                // un-contracted d^2: csrc/nbr.hip pair_test), so the pair set is the one a fresh search at the cutoff finds
                float bx = -dx, by = -dy, bz = -dz;
                if (A.cell.diag) min_image<true>(A.cell, bx, by, bz); else min_image<false>(A.cell, bx, by, bz);
                const float b2 = norm2_ref(bx, by, bz);
                if (!((b2 < tc.rc2) && (b2 != 0.f))) continue;
            }
            apply_shift(A.cell, A.shift[row + k], dx, dy, dz);      // d = x_i - x_j - o.h
            // (the functional form is dispatched HERE, around evaluation AND accumulation: with one run-time switch inside
            //  pair_eval the parameter-derivative slots of PairOut -- filled differently by every form -- merged after the
            //  switch as a private-memory array: 16-20 bytes of scratch per lane in every instantiation)
            const float d2 = dx * dx + dy * dy + dz * dz;
            auto accumulate = [&](const PairOut& o, float ir) {
                vals[0] += 0.5f * o.u;
                if (LEVEL >= 1) {
                    const float rx = dx * ir, ry = dy * ir, rz = dz * ir;
                    gx = fmaf(o.du, rx, gx); gy = fmaf(o.du, ry, gy); gz = fmaf(o.du, rz, gz);
#pragma unroll
                    for (int t = 0; t < MDG_MAX_THETA; ++t)
                        if (t < A.term.n_theta) vals[1 + t] += 0.5f * o.du_dth[t];
                    if (LEVEL >= 2) {
                        const float ax = wxi - A.w[3 * j], ay = wyi - A.w[3 * j + 1], az = wzi - A.w[3 * j + 2];
                        const float a = rx * ax + ry * ay + rz * az;
                        const float c2 = o.d2u * a, c3 = o.du * ir;
                        hx += c2 * rx + c3 * (ax - a * rx);
                        hy += c2 * ry + c3 * (ay - a * ry);
                        hz += c2 * rz + c3 * (az - a * rz);
#pragma unroll
                        for (int t = 0; t < MDG_MAX_THETA; ++t)
                            if (t < A.term.n_theta) vals[1 + MDG_MAX_THETA + t] += 0.5f * o.ddu_dth[t] * a;
                    }
                }
            };
#define MDG_ELL_FORM(KIND_)                                                        \
            case KIND_: {                                                          \
                PairOut o{};                                                       \
                float r, ir;                                                       \
                pair_eval<LEVEL, KIND_>(tc, d2, r, ir, o);                         \
                accumulate(o, ir);                                                 \
            } break;
            switch (tc.kind) {
                MDG_ELL_FORM(MDG_PAIR_LJ) MDG_ELL_FORM(MDG_PAIR_MORSE) MDG_ELL_FORM(MDG_PAIR_BUCK) MDG_ELL_FORM(MDG_PAIR_TABLE)
                default: {
                    PairOut o{};
                    float r, ir;
                    pair_eval<LEVEL, MDG_PAIR_YUKAWA>(tc, d2, r, ir, o);
                    accumulate(o, ir);
                } break;
            }
#undef MDG_ELL_FORM
        }
        if (LEVEL >= 1) {
            gx = group_sum<LPA>(gx); gy = group_sum<LPA>(gy); gz = group_sum<LPA>(gz);
            if (LEVEL >= 2) { hx = group_sum<LPA>(hx); hy = group_sum<LPA>(hy); hz = group_sum<LPA>(hz); }
            if (sub == 0) {
                const float os = A.oscale;
                if (A.grad) {
                    if (A.oacc) { A.grad[3 * i] = fmaf(os, gx, A.grad[3 * i]); A.grad[3 * i + 1] = fmaf(os, gy, A.grad[3 * i + 1]);
                                  A.grad[3 * i + 2] = fmaf(os, gz, A.grad[3 * i + 2]); }
                    else { A.grad[3 * i] = os * gx; A.grad[3 * i + 1] = os * gy; A.grad[3 * i + 2] = os * gz; }
                }
                if (LEVEL >= 2) {
                    if (A.oacc) { A.hw[3 * i] = fmaf(os, hx, A.hw[3 * i]); A.hw[3 * i + 1] = fmaf(os, hy, A.hw[3 * i + 1]);
                                  A.hw[3 * i + 2] = fmaf(os, hz, A.hw[3 * i + 2]); }
                    else { A.hw[3 * i] = os * hx; A.hw[3 * i + 1] = os * hy; A.hw[3 * i + 2] = os * hz; }
                }
            }
        }
    }
    block_sum_n<NSCAL>(vals, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NSCAL; ++k) A.partial[(size_t)blockIdx.x * NSCAL + k] = vals[k];
    }
}

// one wave per scalar, fixed summation order
__global__ void pair_ell_finish(const float* __restrict__ partial, int nblocks, int n_theta,
                                float* energy, float* gtheta, float* gtheta_w) {
    const int k = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int b = lane; b < nblocks; b += 64) s += partial[(size_t)b * NSCAL + k];
    s = wave_sum(s);
    if (lane != 0) return;
    if (k == 0) { if (energy) energy[0] = s; }
    else if (k <= MDG_MAX_THETA) { if (gtheta && k - 1 < n_theta) gtheta[k - 1] = s; }
    else if (gtheta_w && k - 1 - MDG_MAX_THETA < n_theta) gtheta_w[k - 1 - MDG_MAX_THETA] = s;
}

int pick_lpa(int N) {
    int lpa = 64;
    while (lpa > 8 && (long long)N * lpa / 2 >= 131072) lpa /= 2;
    return lpa;
}

}  // namespace

extern "C" int64_t mdg_pair_partial_size(int n_atoms) {
    const int lpa = pick_lpa(n_atoms);
    const int apb = 256 / lpa;
    return (int64_t)((n_atoms + apb - 1) / apb) * NSCAL;
}

#define MDG_ELL_LAUNCH(LPA_)                                                                       \
    case LPA_:                                                                                     \
        if (level == 2) hipLaunchKernelGGL((pair_ell_kernel<LPA_, 2>), grid, dim3(256), 0, st, a); \
        else if (level == 1) hipLaunchKernelGGL((pair_ell_kernel<LPA_, 1>), grid, dim3(256), 0, st, a); \
        else hipLaunchKernelGGL((pair_ell_kernel<LPA_, 0>), grid, dim3(256), 0, st, a);            \
        break;

extern "C" int mdg_pair_eval_ell_into(const float* pos, int n_atoms, const MdgCell* cell, const int32_t* col,
                                      const int32_t* shift, const int32_t* cnt, int max_nbr,
                                      const MdgPairTerm* term, const float* theta, const float* w,
                                      float* energy, float* grad, float* gtheta, float* hw, float* gtheta_w,
                                      float* partial, float out_scale, int accumulate, void* stream);

extern "C" int mdg_pair_eval_ell(const float* pos, int n_atoms, const MdgCell* cell, const int32_t* col,
                                 const int32_t* shift, const int32_t* cnt, int max_nbr,
                                 const MdgPairTerm* term, const float* theta, const float* w,
                                 float* energy, float* grad, float* gtheta, float* hw, float* gtheta_w,
                                 float* partial, void* stream) {
    return mdg_pair_eval_ell_into(pos, n_atoms, cell, col, shift, cnt, max_nbr, term, theta, w, energy, grad, gtheta, hw,
                                  gtheta_w, partial, 1.0f, 0, stream);
}

extern "C" int mdg_pair_eval_ell_into(const float* pos, int n_atoms, const MdgCell* cell, const int32_t* col,
                                      const int32_t* shift, const int32_t* cnt, int max_nbr,
                                      const MdgPairTerm* term, const float* theta, const float* w,
                                      float* energy, float* grad, float* gtheta, float* hw, float* gtheta_w,
                                      float* partial, float out_scale, int accumulate, void* stream) {
    MDG_CHECK_ARG(pos && cell && col && shift && cnt && term && partial, "pair_eval_ell: null buffer");
    MDG_CHECK_ARG(n_atoms > 0 && max_nbr > 0, "pair_eval_ell: bad sizes");
    // (MDG_PAIR_TABLE: theta is the table itself -- 2 p floats -- and carries no parameter gradient here)
    MDG_CHECK_ARG((term->kind >= 0 && term->kind <= MDG_PAIR_YUKAWA && term->n_theta <= MDG_MAX_THETA) ||
                  (term->kind == MDG_PAIR_TABLE && term->p >= 4 && term->n_theta == 2 * term->p && term->phi > 0.f),
                  "pair_eval_ell: bad pair term");
    MDG_CHECK_ARG(term->n_theta == 0 || theta, "pair_eval_ell: theta is null");
    MDG_CHECK_ARG(!w || hw, "pair_eval_ell: w given without hw output");
    const int level = w ? 2 : ((grad || gtheta) ? 1 : 0);
    // accumulate: bit 0 = add onto grad / hw, bit 1 = re-apply the exact cutoff test (Verlet lists)
    EllArgs a{pos, n_atoms, *cell, col, shift, cnt, max_nbr, *term, theta, w, grad, hw, partial, out_scale, accumulate & 1,
              (accumulate >> 1) & 1};
    const int lpa = pick_lpa(n_atoms);
    const int apb = 256 / lpa;
    const int nblocks = (n_atoms + apb - 1) / apb;
    dim3 grid(nblocks);
    hipStream_t st = (hipStream_t)stream;
    switch (lpa) { MDG_ELL_LAUNCH(8) MDG_ELL_LAUNCH(16) MDG_ELL_LAUNCH(32) MDG_ELL_LAUNCH(64) }
    if (energy || gtheta || gtheta_w)            // (a force-only evaluation has no scalar to finish)
        hipLaunchKernelGGL(pair_ell_finish, dim3(NSCAL), dim3(64), 0, st, partial, nblocks, term->n_theta, energy,
                           gtheta, gtheta_w);
    MDG_CHECK_LAUNCH("pair_ell_kernel");
    return MDG_OK;
}
