// Bonded terms over a static topology table (torchmd/interface.py:406-510, the polymer demo's
// Stack at demo/fold.py:131-161): energy, force -dU/dx and the Hessian-vector product H w the adjoint needs
// (what double autograd derives at torchmd/sovlers.py:229-233), ONE launch, no atomics.
//
//   bond  (interface.py:447-455)  U = 1/2 k (|b|^2 - ro)^2          b  = x_i - x_j + o L       -- SQUARED length vs ro
//   angle (interface.py:496-508)  U = 1/2 k (theta - theta0)^2      b1 = x_i - x_j + o1 L, b2 = x_k - x_j + o2 L,
//                                 theta = acos( b1.b2 / sqrt(|b1|^2 |b2|^2) )
//   image flags (topology.py:75-80, NON-strict on the upper side):  o = -[b >= L/2] + [b < -L/2]  per component,
//   L = diagonal of the cell (both classes take cell.diag()); o is piecewise constant, so it carries no derivative.
//
// Atom-centric: thread n walks the incidence list of atom n (entries 4 term + role, sorted by term: a fixed summation
// order) and re-derives each term it takes part in -- a chain atom sits in <= 2 bonds / <= 3 angles, cheaper than a
// term-wise pass plus a scatter, and deterministic.  H w comes from forward-mode (dual-number) differentiation of the
// gradient along (w_i - w_j [, w_k - w_j]): exact, no hand-written second derivatives of acos.
#include "common.hpp"

namespace {

struct Dual {
    float v, d;
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, fmaf(a.d, b.v, a.v * b.d)}; }
__device__ __forceinline__ Dual operator*(float a, Dual b) { return {a * b.v, a * b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, float b) { return {a.v - b, a.d}; }
__device__ __forceinline__ Dual operator-(float a, Dual b) { return {a - b.v, -b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
    const float q = a.v / b.v;
    return {q, (a.d - q * b.d) / b.v};
}
__device__ __forceinline__ Dual dsqrt(Dual a) {
    const float s = sqrtf(a.v);
    return {s, 0.5f * a.d / s};
}
__device__ __forceinline__ Dual dacos(Dual a) { return {acosf(a.v), -a.d / sqrtf(1.f - a.v * a.v)}; }

__device__ __forceinline__ float fsqrt_(float a) { return sqrtf(a); }
__device__ __forceinline__ Dual fsqrt_(Dual a) { return dsqrt(a); }
__device__ __forceinline__ float facos_(float a) { return acosf(a); }
__device__ __forceinline__ Dual facos_(Dual a) { return dacos(a); }
__device__ __forceinline__ float val(float a) { return a; }
__device__ __forceinline__ float val(Dual a) { return a.v; }

// topology.get_offsets (topology.py:75-80): -[b >= L/2] + [b < -L/2]
__device__ __forceinline__ float image_flag(float b, float L) { return (b < -0.5f * L ? 1.f : 0.f) - (b >= 0.5f * L ? 1.f : 0.f); }

struct BondedArgs {
    const float* pos;
    const float* w;
    const int32_t* top;
    const int32_t* inc_ptr;
    const int32_t* inc;
    float* e_atom;
    float* grad;
    float* hw;
    int n_atoms;
    float L[3], k, x0, scale;
    int accumulate;
};

// dU/db of the bond term; T = float (value) or Dual (value + directional derivative)
template <typename T>
__device__ __forceinline__ void bond_grad(const T (&b)[3], float k, float ro, T (&g)[3], float& U) {
    const T s = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];                  // interface.py:452
    const T c = (2.f * k) * (s - ro);
    U = 0.5f * k * (val(s) - ro) * (val(s) - ro);
#pragma unroll
    for (int a = 0; a < 3; ++a) g[a] = c * b[a];
}

// dU/db1, dU/db2 of the angle term
template <typename T>
__device__ __forceinline__ void angle_grad(const T (&b1)[3], const T (&b2)[3], float k, float th0, T (&g1)[3], T (&g2)[3], float& U) {
    const T dot = b1[0] * b2[0] + b1[1] * b2[1] + b1[2] * b2[2];         // interface.py:500
    const T n1 = b1[0] * b1[0] + b1[1] * b1[1] + b1[2] * b1[2];
    const T n2 = b2[0] * b2[0] + b2[1] * b2[1] + b2[2] * b2[2];
    const T nrm = fsqrt_(n1 * n2);                                         // :501
    const T cs = dot / nrm;                                                // :503
    const T th = facos_(cs);                                               // :505
    const float dth = val(th) - th0;
    U = 0.5f * k * dth * dth;                                              // :507
    // dU/dcos = k (theta - theta0) * (-1 / sqrt(1 - cos^2)) ;  dcos/db1 = b2 / nrm - cos b1 / n1
    const T one_m = 1.f - cs * cs;
    const T dUdc = (-k) * ((th - th0) / fsqrt_(one_m));
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        g1[a] = dUdc * (b2[a] / nrm - cs * (b1[a] / n1));
        g2[a] = dUdc * (b1[a] / nrm - cs * (b2[a] / n2));
    }
}

template <int KIND, bool HVP>
__global__ __launch_bounds__(256) void bonded_kernel(const BondedArgs A) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= A.n_atoms) return;
    float gx = 0.f, gy = 0.f, gz = 0.f, hx = 0.f, hy = 0.f, hz = 0.f, e = 0.f;
    const int lo = A.inc_ptr[n], hi = A.inc_ptr[n + 1];
    for (int u = lo; u < hi; ++u) {
        const int code = A.inc[u], t = code >> 2, role = code & 3;
        if (KIND == 0) {
            const int i = A.top[2 * t], j = A.top[2 * t + 1];
            float b[3], db[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float r = A.pos[3 * i + a] - A.pos[3 * j + a];
                b[a] = fmaf(image_flag(r, A.L[a]), A.L[a], r);
                if (HVP) db[a] = A.w[3 * i + a] - A.w[3 * j + a];
            }
            const float sg = role == 0 ? 1.f : -1.f;
            float U;
            if (HVP) {
                const Dual bd[3] = {{b[0], db[0]}, {b[1], db[1]}, {b[2], db[2]}};
                Dual g[3];
                bond_grad<Dual>(bd, A.k, A.x0, g, U);
                gx = fmaf(sg, g[0].v, gx); gy = fmaf(sg, g[1].v, gy); gz = fmaf(sg, g[2].v, gz);
                hx = fmaf(sg, g[0].d, hx); hy = fmaf(sg, g[1].d, hy); hz = fmaf(sg, g[2].d, hz);
            } else {
                float g[3];
                bond_grad<float>(b, A.k, A.x0, g, U);
                gx = fmaf(sg, g[0], gx); gy = fmaf(sg, g[1], gy); gz = fmaf(sg, g[2], gz);
            }
            if (role == 0) e += U;
        } else {
            const int i = A.top[3 * t], j = A.top[3 * t + 1], kk = A.top[3 * t + 2];
            float b1[3], b2[3], d1[3] = {0.f, 0.f, 0.f}, d2[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float xj = A.pos[3 * j + a];
                const float r1 = A.pos[3 * i + a] - xj, r2 = A.pos[3 * kk + a] - xj;
                b1[a] = fmaf(image_flag(r1, A.L[a]), A.L[a], r1);
                b2[a] = fmaf(image_flag(r2, A.L[a]), A.L[a], r2);
                if (HVP) {
                    const float wj = A.w[3 * j + a];
                    d1[a] = A.w[3 * i + a] - wj; d2[a] = A.w[3 * kk + a] - wj;
                }
            }
            // role 0: dU/dx_i = g1 ; role 2: dU/dx_k = g2 ; role 1 (the centre): -(g1 + g2)
            const float s1 = role == 0 ? 1.f : (role == 1 ? -1.f : 0.f), s2 = role == 2 ? 1.f : (role == 1 ? -1.f : 0.f);
            float U;
            if (HVP) {
                const Dual b1d[3] = {{b1[0], d1[0]}, {b1[1], d1[1]}, {b1[2], d1[2]}};
                const Dual b2d[3] = {{b2[0], d2[0]}, {b2[1], d2[1]}, {b2[2], d2[2]}};
                Dual g1[3], g2[3];
                angle_grad<Dual>(b1d, b2d, A.k, A.x0, g1, g2, U);
                gx += s1 * g1[0].v + s2 * g2[0].v; gy += s1 * g1[1].v + s2 * g2[1].v; gz += s1 * g1[2].v + s2 * g2[2].v;
                hx += s1 * g1[0].d + s2 * g2[0].d; hy += s1 * g1[1].d + s2 * g2[1].d; hz += s1 * g1[2].d + s2 * g2[2].d;
            } else {
                float g1[3], g2[3];
                angle_grad<float>(b1, b2, A.k, A.x0, g1, g2, U);
                gx += s1 * g1[0] + s2 * g2[0]; gy += s1 * g1[1] + s2 * g2[1]; gz += s1 * g1[2] + s2 * g2[2];
            }
            if (role == 0) e += U;
        }
    }
    if (A.e_atom) A.e_atom[n] = e;
    if (A.grad) {
        float* o = A.grad + 3 * n;
        const float s = A.scale;
        if (A.accumulate) { o[0] = fmaf(s, gx, o[0]); o[1] = fmaf(s, gy, o[1]); o[2] = fmaf(s, gz, o[2]); }
        else { o[0] = s * gx; o[1] = s * gy; o[2] = s * gz; }
    }
    if (HVP && A.hw) {
        float* o = A.hw + 3 * n;
        const float s = A.scale;
        if (A.accumulate) { o[0] = fmaf(s, hx, o[0]); o[1] = fmaf(s, hy, o[1]); o[2] = fmaf(s, hz, o[2]); }
        else { o[0] = s * hx; o[1] = s * hy; o[2] = s * hz; }
    }
}

}  // namespace

extern "C" int mdg_bonded_eval(const float* pos, int n_atoms, const float* cell_len, int kind, const int32_t* top, int n_terms,
                               float k, float x0, const int32_t* inc_ptr, const int32_t* inc, const float* w, float* e_atom,
                               float* grad, float* hw, float out_scale, int accumulate, void* stream) {
    MDG_CHECK_ARG(pos && cell_len && n_atoms > 0, "bonded_eval: bad arguments");
    MDG_CHECK_ARG(kind == MDG_BONDED_BOND || kind == MDG_BONDED_ANGLE, "bonded_eval: kind must be MDG_BONDED_BOND or MDG_BONDED_ANGLE");
    MDG_CHECK_ARG(n_terms >= 0 && inc_ptr && (n_terms == 0 || (top && inc)), "bonded_eval: topology table missing");
    MDG_CHECK_ARG(!hw || w, "bonded_eval: the Hessian-vector product needs w");
    MDG_CHECK_ARG(e_atom || grad || hw, "bonded_eval: no output requested");
    BondedArgs a{pos, w, top, inc_ptr, inc, e_atom, grad, hw, n_atoms, {cell_len[0], cell_len[1], cell_len[2]}, k, x0,
                 out_scale, accumulate};
    const dim3 grid((n_atoms + 255) / 256), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (kind == MDG_BONDED_BOND) {
        if (hw) hipLaunchKernelGGL((bonded_kernel<0, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((bonded_kernel<0, false>), grid, block, 0, st, a);
    } else {
        if (hw) hipLaunchKernelGGL((bonded_kernel<1, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((bonded_kernel<1, false>), grid, block, 0, st, a);
    }
    MDG_CHECK_LAUNCH("bonded_kernel");
    return MDG_OK;
}
