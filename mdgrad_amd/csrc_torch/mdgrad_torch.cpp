// torch.ops.mdgrad.* -- the TORCH_LIBRARY op layer of SURVEY 8b above the C ABI of libmdgrad_hip.so
// (include/mdgrad_hip.h).  Host-only C++: every op validates its tensors (TORCH_CHECK -> Python RuntimeError),
// allocates its outputs from the caching allocator, and enqueues the kernels of the C entry point on the current
// HIP stream of the tensors' device.  Stateless and re-entrant; no host synchronisation.  The differentiation
// contract stays in the Python autograd.Functions (mdgrad_amd/ops.py), which call these ops instead of the ctypes
// bindings when this library is present (≈ 3x less host time per launch).
//
//   nbr_build        K1   torchmd/topology.py:30-73        -> mdg_nbr_build_dense(_groups) / mdg_nbr_build_cell(_groups)
//   pair_force       K2/3 interface.py:298-300 + autograd  -> mdg_pair_eval_ell (energy, dU/dx, dU/dtheta)
//   pair_hvp         K4   sovlers.py:229-233 (double bwd)   -> mdg_pair_eval_ell (H w, d(w.dU/dx)/dtheta)
//   nhc_vv_forward   K5/6 sovlers.py:106-127 / :21-40       -> mdg_traj_fwd_small
//   nhc_vv_adjoint   K7   sovlers.py:211-293                -> mdg_traj_adj_small
//   rdf_fwd/rdf_bwd  K8   observable.py:62-76               -> mdg_rdf_fwd_uniform / mdg_rdf_bwd_uniform
//   edge_geom(+_bwd) schnet.py:142                          -> mdg_edge_geom / mdg_edge_geom_bwd
//   cfconv_fwd/_bwd  K9+K10 modules.py:531-571              -> mdg_cfconv_fwd(_bf16) / mdg_cfconv_bwd(_bf16)
//   dense_ssp        K11/12 layers.py:86-134                -> mdg_dense
//   ssp_dual_bwd_t, atb                                     -> mdg_ssp_dual_bwd_t / mdg_atb
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <tuple>
#include <vector>

#include "../../include/mdgrad_hip.h"

namespace {

using at::Tensor;
using OptTensor = c10::optional<Tensor>;

void* stream_of(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

void check_f32(const Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda(), "mdgrad: ", name, " must live on a HIP device");
    TORCH_CHECK(t.scalar_type() == at::kFloat, "mdgrad: ", name, " must be float32");
    TORCH_CHECK(t.is_contiguous(), "mdgrad: ", name, " must be contiguous");
}
void check_i32(const Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kInt && t.is_contiguous(), "mdgrad: ", name,
                " must be a contiguous int32 tensor on a HIP device");
}
void same_device(const Tensor& a, const Tensor& b, const char* name) {
    TORCH_CHECK(a.device() == b.device(), "mdgrad: ", name, " is on a different device");
}
const float* fptr(const Tensor& t) { return t.data_ptr<float>(); }
const float* fptr(const OptTensor& t, const char* name) {
    if (!t.has_value() || !t->defined()) return nullptr;
    check_f32(*t, name);
    return t->data_ptr<float>();
}
float* mptr(Tensor& t) { return t.data_ptr<float>(); }
float* mptr(OptTensor& t) { return (t.has_value() && t->defined()) ? t->data_ptr<float>() : nullptr; }
void ok(int rc) { TORCH_CHECK(rc == 0, "mdgrad: ", mdg_last_error()); }
// bf16 mirrors of node matrices (the rows16 kernels, include/mdgrad_hip.h)
bool is_bf16(const Tensor& t) { return t.scalar_type() == at::kBFloat16; }
const uint16_t* hptr(const Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda() && is_bf16(t) && t.is_contiguous(), "mdgrad: ", name, " must be a contiguous bfloat16 tensor on a HIP device");
    return reinterpret_cast<const uint16_t*>(t.data_ptr());
}
const uint16_t* hptr(const OptTensor& t, const char* name) { return (t.has_value() && t->defined()) ? hptr(*t, name) : nullptr; }

// cell = the 9 row-major entries of h followed by the 9 of its inverse (the Python side computes the inverse the way
// the reference does, topology.py:59) and a diagonal flag
MdgCell make_cell(at::ArrayRef<double> c) {
    TORCH_CHECK(c.size() == 19, "mdgrad: cell descriptor = 9 (h) + 9 (inverse) + 1 (diag flag) numbers");
    MdgCell m;
    for (int k = 0; k < 9; ++k) { m.h[k] = (float)c[k]; m.inv[k] = (float)c[9 + k]; }
    m.diag = c[18] != 0.0;
    return m;
}

// one pair term = ints (kind, p, q, theta_off, n_theta) + floats (c, a, phi, cutoff) (+ optional [N,N] uint8 mask)
MdgPairTerm make_term(at::ArrayRef<int64_t> ti, at::ArrayRef<double> tf, const OptTensor& mask) {
    TORCH_CHECK(ti.size() == 5 && tf.size() == 4, "mdgrad: pair term = 5 ints (kind, p, q, theta_off, n_theta) + 4 floats "
                "(c, a, phi, cutoff)");
    MdgPairTerm t{};
    t.kind = (int32_t)ti[0]; t.p = (int32_t)ti[1]; t.q = (int32_t)ti[2]; t.theta_off = (int32_t)ti[3]; t.n_theta = (int32_t)ti[4];
    t.c = (float)tf[0]; t.a = (float)tf[1]; t.phi = (float)tf[2]; t.cutoff = (float)tf[3];
    t.mask = nullptr;
    if (mask.has_value() && mask->defined()) {
        TORCH_CHECK(mask->is_cuda() && mask->scalar_type() == at::kByte && mask->is_contiguous(), "mdgrad: mask must be a "
                    "contiguous uint8 tensor on the device");
        t.mask = mask->data_ptr<uint8_t>();
    }
    return t;
}

// ------------------------------------------------------------------------------------------------ K1
// (col, shift, cnt, overflow): the padded per-atom full list of include/mdgrad_hip.h; `overflow` stays on the device
std::tuple<Tensor, Tensor, Tensor, Tensor> nbr_build(const Tensor& pos, at::ArrayRef<double> cell, double cutoff,
                                                     const OptTensor& mask, int64_t max_nbr, int64_t group, bool cell_list) {
    check_f32(pos, "pos");
    TORCH_CHECK(pos.dim() == 2 && pos.size(1) == 3, "mdgrad: pos must be [N,3]");
    const MdgCell c = make_cell(cell);
    const int64_t n = pos.size(0);
    const auto io = pos.options().dtype(at::kInt);
    Tensor col = at::empty({n, max_nbr}, io), shift = at::empty({n, max_nbr}, io), cnt = at::empty({n}, io);
    Tensor overflow = at::zeros({1}, io);
    const uint8_t* mk = nullptr;
    if (mask.has_value() && mask->defined()) {
        TORCH_CHECK(mask->is_cuda() && mask->scalar_type() == at::kByte && mask->is_contiguous(), "mdgrad: mask must be uint8");
        mk = mask->data_ptr<uint8_t>();
    }
    const int g = (int)(group > 0 ? group : n);
    void* st = stream_of(pos);
    if (cell_list) {
        const int64_t words = mdg_nbr_cell_scratch_groups((int)n, g, &c, (float)cutoff);
        TORCH_CHECK(words > 0, "mdgrad: the cell list needs an orthorhombic cell with at least 3 bins per side");
        Tensor scratch = at::empty({words}, io);
        ok(mdg_nbr_build_cell_groups(fptr(pos), (int)n, g, &c, (float)cutoff, mk, col.data_ptr<int32_t>(),
                                     shift.data_ptr<int32_t>(), cnt.data_ptr<int32_t>(), (int)max_nbr,
                                     overflow.data_ptr<int32_t>(), scratch.data_ptr<int32_t>(), st));
    } else {
        ok(mdg_nbr_build_dense_groups(fptr(pos), (int)n, g, &c, (float)cutoff, mk, col.data_ptr<int32_t>(),
                                      shift.data_ptr<int32_t>(), cnt.data_ptr<int32_t>(), (int)max_nbr,
                                      overflow.data_ptr<int32_t>(), st));
    }
    return {col, shift, cnt, overflow};
}

// ------------------------------------------------------------------------------------------------ K2-K4
struct EllRef { const int32_t* col; const int32_t* shift; const int32_t* cnt; int max_nbr; };
EllRef ell_of(const Tensor& pos, const Tensor& col, const Tensor& shift, const Tensor& cnt) {
    check_i32(col, "col"); check_i32(shift, "shift"); check_i32(cnt, "cnt");
    same_device(pos, col, "col");
    TORCH_CHECK(col.dim() == 2 && col.size(0) == pos.size(0) && shift.sizes() == col.sizes() && cnt.numel() == pos.size(0),
                "mdgrad: neighbour list does not match pos");
    return EllRef{col.data_ptr<int32_t>(), shift.data_ptr<int32_t>(), cnt.data_ptr<int32_t>(), (int)col.size(1)};
}

// (U[1], dU/dx [N,3], dU/dtheta [K])
std::tuple<Tensor, Tensor, Tensor> pair_force(const Tensor& pos, at::ArrayRef<double> cell, const Tensor& col,
                                              const Tensor& shift, const Tensor& cnt, at::ArrayRef<int64_t> term_i,
                                              at::ArrayRef<double> term_f, const OptTensor& mask, const OptTensor& theta) {
    check_f32(pos, "pos");
    const MdgCell c = make_cell(cell);
    const MdgPairTerm t = make_term(term_i, term_f, mask);
    const EllRef e = ell_of(pos, col, shift, cnt);
    const int n = (int)pos.size(0);
    Tensor U = at::empty({1}, pos.options()), g = at::empty_like(pos);
    Tensor gth = at::zeros({std::max<int64_t>(1, theta.has_value() && theta->defined() ? theta->numel() : 0)}, pos.options());
    Tensor partial = at::empty({mdg_pair_partial_size(n)}, pos.options());
    ok(mdg_pair_eval_ell(fptr(pos), n, &c, e.col, e.shift, e.cnt, e.max_nbr, &t, fptr(theta, "theta"), nullptr, mptr(U), mptr(g),
                         t.n_theta ? mptr(gth) : nullptr, nullptr, nullptr, mptr(partial), stream_of(pos)));
    return {U, g, gth};
}

// (H w [N,3], d(w . dU/dx)/dtheta [K])
std::tuple<Tensor, Tensor> pair_hvp(const Tensor& pos, at::ArrayRef<double> cell, const Tensor& col, const Tensor& shift,
                                    const Tensor& cnt, at::ArrayRef<int64_t> term_i, at::ArrayRef<double> term_f,
                                    const OptTensor& mask, const OptTensor& theta, const Tensor& w) {
    check_f32(pos, "pos"); check_f32(w, "w");
    TORCH_CHECK(w.sizes() == pos.sizes(), "mdgrad: w must have the shape of pos");
    const MdgCell c = make_cell(cell);
    const MdgPairTerm t = make_term(term_i, term_f, mask);
    const EllRef e = ell_of(pos, col, shift, cnt);
    const int n = (int)pos.size(0);
    Tensor hw = at::empty_like(pos);
    Tensor gth = at::zeros({std::max<int64_t>(1, theta.has_value() && theta->defined() ? theta->numel() : 0)}, pos.options());
    Tensor partial = at::empty({mdg_pair_partial_size(n)}, pos.options());
    ok(mdg_pair_eval_ell(fptr(pos), n, &c, e.col, e.shift, e.cnt, e.max_nbr, &t, fptr(theta, "theta"), fptr(w), nullptr, nullptr,
                         nullptr, mptr(hw), t.n_theta ? mptr(gth) : nullptr, mptr(partial), stream_of(pos)));
    return {hw, gth};
}

// ------------------------------------------------------------------------------------------------ K5-K7
struct TrajDesc { MdgTrajParams prm; MdgCell cell; MdgTerms terms; };
// iprm = (n_rep, n_atoms, n_frames, n_chains, ensemble, block); fprm = (T, n_dof, Q[0..n_chains));
// terms_i / terms_f = the terms' 5 ints / 4 floats back to back (no masks: masked Stacks take the generic path)
TrajDesc traj_desc(at::ArrayRef<int64_t> iprm, at::ArrayRef<double> fprm, at::ArrayRef<double> cell,
                   at::ArrayRef<int64_t> terms_i, at::ArrayRef<double> terms_f, int64_t n_theta_total) {
    TORCH_CHECK(iprm.size() == 6, "mdgrad: iprm = (n_rep, n_atoms, n_frames, n_chains, ensemble, block)");
    TrajDesc d{};
    d.prm.n_rep = (int32_t)iprm[0]; d.prm.n_atoms = (int32_t)iprm[1]; d.prm.n_frames = (int32_t)iprm[2];
    d.prm.n_chains = (int32_t)iprm[3]; d.prm.ensemble = (int32_t)iprm[4]; d.prm.block = (int32_t)iprm[5];
    TORCH_CHECK(fprm.size() >= 2 && fprm.size() <= 2 + MDG_MAX_CHAINS, "mdgrad: fprm = (T, n_dof, Q...)");
    d.prm.T = (float)fprm[0]; d.prm.n_dof = (float)fprm[1];
    for (size_t k = 2; k < fprm.size(); ++k) d.prm.Q[k - 2] = (float)fprm[k];
    d.cell = make_cell(cell);
    const size_t nt = terms_i.size() / 5;
    TORCH_CHECK(nt >= 1 && nt <= MDG_MAX_TERMS && terms_i.size() == 5 * nt && terms_f.size() == 4 * nt,
                "mdgrad: 1..", MDG_MAX_TERMS, " pair terms of 5 ints + 4 floats");
    d.terms.n_terms = (int32_t)nt; d.terms.n_theta_total = (int32_t)n_theta_total;
    for (size_t m = 0; m < nt; ++m)
        d.terms.t[m] = make_term(terms_i.slice(5 * m, 5), terms_f.slice(4 * m, 4), OptTensor());
    return d;
}

// (v_t [R,T,N,3], q_t [R,T,N,3], pv_t [R,T,C] (empty for NVE), nonfinite int32[R])
std::tuple<Tensor, Tensor, Tensor, Tensor> nhc_vv_forward(const Tensor& v0, const Tensor& q0, const OptTensor& pv0,
                                                          const Tensor& mass, const Tensor& t, const OptTensor& theta,
                                                          at::ArrayRef<int64_t> iprm, at::ArrayRef<double> fprm,
                                                          at::ArrayRef<double> cell, at::ArrayRef<int64_t> terms_i,
                                                          at::ArrayRef<double> terms_f, int64_t n_theta_total) {
    check_f32(v0, "v0"); check_f32(q0, "q0"); check_f32(mass, "mass"); check_f32(t, "t");
    const TrajDesc d = traj_desc(iprm, fprm, cell, terms_i, terms_f, n_theta_total);
    const int64_t R = d.prm.n_rep, T = d.prm.n_frames, N = d.prm.n_atoms, C = d.prm.n_chains;
    TORCH_CHECK(v0.numel() == R * N * 3 && q0.numel() == R * N * 3 && t.numel() == T, "mdgrad: state / time grid do not match iprm");
    const bool nhc = d.prm.ensemble == 0;
    Tensor v_t = at::empty({R, T, N, 3}, v0.options()), q_t = at::empty({R, T, N, 3}, v0.options());
    Tensor pv_t = at::empty({nhc ? R : 0, nhc ? T : 0, nhc ? C : 0}, v0.options());
    Tensor bad = at::zeros({R}, v0.options().dtype(at::kInt));
    ok(mdg_traj_fwd_small(&d.prm, &d.cell, &d.terms, fptr(theta, "theta"), fptr(mass), fptr(t), fptr(v0), fptr(q0),
                          fptr(pv0, "pv0"), mptr(v_t), mptr(q_t), nhc ? mptr(pv_t) : nullptr, bad.data_ptr<int32_t>(),
                          stream_of(v0)));
    return {v_t, q_t, pv_t, bad};
}

// (adj_v0 [R,N,3], adj_q0 [R,N,3], adj_pv0 [R,C], adj_theta [R,K])
std::tuple<Tensor, Tensor, Tensor, Tensor> nhc_vv_adjoint(const Tensor& v_t, const Tensor& q_t, const OptTensor& pv_t,
                                                          const OptTensor& g_v, const OptTensor& g_q, const OptTensor& g_pv,
                                                          const Tensor& mass, const Tensor& t, const OptTensor& theta,
                                                          at::ArrayRef<int64_t> iprm, at::ArrayRef<double> fprm,
                                                          at::ArrayRef<double> cell, at::ArrayRef<int64_t> terms_i,
                                                          at::ArrayRef<double> terms_f, int64_t n_theta_total) {
    check_f32(v_t, "v_t"); check_f32(q_t, "q_t"); check_f32(mass, "mass"); check_f32(t, "t");
    const TrajDesc d = traj_desc(iprm, fprm, cell, terms_i, terms_f, n_theta_total);
    const int64_t R = d.prm.n_rep, T = d.prm.n_frames, N = d.prm.n_atoms, C = d.prm.n_chains;
    TORCH_CHECK(v_t.numel() == R * T * N * 3 && q_t.numel() == R * T * N * 3, "mdgrad: trajectories do not match iprm");
    const bool nhc = d.prm.ensemble == 0;
    Tensor av = at::empty({R, N, 3}, v_t.options()), aq = at::empty({R, N, 3}, v_t.options());
    Tensor ap = at::empty({nhc ? R : 0, nhc ? C : 0}, v_t.options());
    Tensor ath = at::zeros({R, std::max<int64_t>(1, n_theta_total)}, v_t.options());
    ok(mdg_traj_adj_small(&d.prm, &d.cell, &d.terms, fptr(theta, "theta"), fptr(mass), fptr(t), fptr(v_t), fptr(q_t),
                          fptr(pv_t, "pv_t"), fptr(g_v, "g_v"), fptr(g_q, "g_q"), fptr(g_pv, "g_pv"), mptr(av), mptr(aq),
                          nhc ? mptr(ap) : nullptr, n_theta_total ? mptr(ath) : nullptr, stream_of(v_t)));
    return {av, aq, ap, ath};
}

// ------------------------------------------------------------------------------------------------ K8
Tensor rdf_fwd(const Tensor& xyz, at::ArrayRef<double> cell, double cutoff, const OptTensor& mask, const Tensor& mu,
               double spacing, double coeff) {
    check_f32(xyz, "xyz"); check_f32(mu, "mu");
    TORCH_CHECK(xyz.dim() == 3 && xyz.size(2) == 3, "mdgrad: xyz must be [F,N,3]");
    const MdgCell c = make_cell(cell);
    const int F = (int)xyz.size(0), N = (int)xyz.size(1), B = (int)mu.numel();
    Tensor raw = at::empty({B}, xyz.options()), partial = at::empty({mdg_rdf_partial_size(F, N, B)}, xyz.options());
    const uint8_t* mk = (mask.has_value() && mask->defined()) ? mask->data_ptr<uint8_t>() : nullptr;
    ok(mdg_rdf_fwd_uniform(fptr(xyz), F, N, &c, (float)cutoff, mk, fptr(mu), (float)spacing, (float)coeff, B, mptr(raw),
                           mptr(partial), stream_of(xyz)));
    return raw;
}
Tensor rdf_bwd(const Tensor& xyz, at::ArrayRef<double> cell, double cutoff, const OptTensor& mask, const Tensor& mu,
               double spacing, double coeff, const Tensor& g_raw) {
    check_f32(xyz, "xyz"); check_f32(mu, "mu"); check_f32(g_raw, "g_raw");
    const MdgCell c = make_cell(cell);
    const int F = (int)xyz.size(0), N = (int)xyz.size(1), B = (int)mu.numel();
    TORCH_CHECK(g_raw.numel() == B, "mdgrad: g_raw must have one entry per centre");
    Tensor g = at::empty_like(xyz);
    const uint8_t* mk = (mask.has_value() && mask->defined()) ? mask->data_ptr<uint8_t>() : nullptr;
    ok(mdg_rdf_bwd_uniform(fptr(xyz), F, N, &c, (float)cutoff, mk, fptr(mu), (float)spacing, (float)coeff, B, fptr(g_raw),
                           mptr(g), stream_of(xyz)));
    return g;
}

// ------------------------------------------------------------------------------------------------ SchNet block
MdgFilterNet filter_net(const Tensor& mu, const Tensor& coef, const Tensor& W1, const Tensor& b1, const Tensor& W2,
                        const Tensor& b2) {
    check_f32(mu, "mu"); check_f32(coef, "coef"); check_f32(W1, "W1"); check_f32(b1, "b1"); check_f32(W2, "W2"); check_f32(b2, "b2");
    MdgFilterNet n{};
    n.mu = fptr(mu); n.coef = fptr(coef); n.W1 = fptr(W1); n.b1 = fptr(b1); n.W2 = fptr(W2); n.b2 = fptr(b2);
    n.n_gauss = (int32_t)mu.numel(); n.n_filters = (int32_t)W2.size(0);
    TORCH_CHECK(W1.dim() == 2 && W1.size(0) == n.n_gauss && W1.size(1) == n.n_gauss && W2.dim() == 2 && W2.size(1) == n.n_gauss,
                "mdgrad: filter network shapes (W1 [G,G], W2 [F,G])");
    TORCH_CHECK(mdg_cfconv_supported(n.n_gauss, n.n_filters), "mdgrad: fused cfconv needs G <= 64 and F a multiple of 4 up to 64, of 8 up to 128, or of 128 up to 512");
    return n;
}

// (d, uhat, dd, ddel): dd / ddel are empty without w
std::tuple<Tensor, Tensor, Tensor, Tensor> edge_geom(const Tensor& x, const OptTensor& w, const Tensor& nbr, const Tensor& offsets) {
    check_f32(x, "x"); check_f32(offsets, "offsets");
    TORCH_CHECK(nbr.is_cuda() && nbr.scalar_type() == at::kLong && nbr.is_contiguous() && nbr.dim() == 2 && nbr.size(1) == 2,
                "mdgrad: nbr must be a contiguous int64 [E,2] tensor on the device");
    const int64_t E = nbr.size(0);
    const bool tan = w.has_value() && w->defined();
    Tensor d = at::empty({E}, x.options()), u = at::empty({E, 3}, x.options());
    Tensor dd = at::empty({tan ? E : 0}, x.options()), ddel = at::empty({tan ? E : 0, 3}, x.options());
    ok(mdg_edge_geom(fptr(x), fptr(w, "w"), nbr.data_ptr<int64_t>(), fptr(offsets), E, mptr(d), mptr(u), tan ? mptr(dd) : nullptr,
                     tan ? mptr(ddel) : nullptr, stream_of(x)));
    return {d, u, dd, ddel};
}

// (force [N,3], dwf [N,3] (empty without d_b))
std::tuple<Tensor, Tensor> edge_geom_bwd(const OptTensor& d_b, const Tensor& dd_b, const OptTensor& d, const OptTensor& dd,
                                         const Tensor& uhat, const OptTensor& ddel, const Tensor& col, const Tensor& eid,
                                         const Tensor& cnt) {
    check_f32(dd_b, "dd_b"); check_f32(uhat, "uhat");
    check_i32(col, "col"); check_i32(eid, "eid"); check_i32(cnt, "cnt");
    const int64_t N = cnt.numel();
    const bool full = d_b.has_value() && d_b->defined();
    TORCH_CHECK(!full || (d.has_value() && d->defined()), "mdgrad: d(w.F)/dx needs the distances d");
    Tensor f = at::empty({N, 3}, uhat.options()), dwf = at::empty({full ? N : 0, 3}, uhat.options());
    ok(mdg_edge_geom_bwd(fptr(d_b, "d_b"), fptr(dd_b), fptr(d, "d"), fptr(dd, "dd"), fptr(uhat), fptr(ddel, "ddel"),
                         col.data_ptr<int32_t>(), eid.data_ptr<int32_t>(), cnt.data_ptr<int32_t>(), (int)N, (int)col.size(1), mptr(f),
                         full ? mptr(dwf) : nullptr, stream_of(uhat)));
    return {f, dwf};
}

// (m, md, hsum, hdsum); md / hsum / hdsum are empty when not requested
std::tuple<Tensor, Tensor, Tensor, Tensor> cfconv_fwd(const Tensor& mu, const Tensor& coef, const Tensor& W1, const Tensor& b1,
                                                      const Tensor& W2, const Tensor& b2, bool bf16, const Tensor& d,
                                                      const OptTensor& dd, const Tensor& h, const OptTensor& hd,
                                                      const Tensor& col, const Tensor& eid, const Tensor& cnt, bool want_sums) {
    const MdgFilterNet net = filter_net(mu, coef, W1, b1, W2, b2);
    const bool r16 = is_bf16(h);                 // bf16 mirrors of the node rows: mdg_cfconv_fwd_rows16
    check_f32(d, "d");
    if (!r16) check_f32(h, "h");
    check_i32(col, "col"); check_i32(eid, "eid"); check_i32(cnt, "cnt");
    const int64_t N = cnt.numel(), F = net.n_filters;
    TORCH_CHECK(h.dim() == 2 && h.size(0) == N && h.size(1) == F, "mdgrad: h must be [N,F]");
    const bool tan = dd.has_value() && dd->defined(), htan = hd.has_value() && hd->defined();
    const auto fo = h.options().dtype(at::kFloat);
    Tensor m = at::empty({N, F}, fo), md = at::empty({tan ? N : 0, F}, fo);
    Tensor hs = at::empty({want_sums ? N : 0, F}, fo), hds = at::empty({want_sums && htan ? N : 0, F}, fo);
    if (r16) {
        TORCH_CHECK(bf16, "mdgrad: bf16 node rows go with the bf16 filter kernels");
        ok(mdg_cfconv_fwd_rows16(&net, fptr(d), fptr(dd, "dd"), hptr(h, "h"), hptr(hd, "hd"), col.data_ptr<int32_t>(),
                                 eid.data_ptr<int32_t>(), cnt.data_ptr<int32_t>(), (int)N, (int)col.size(1), mptr(m),
                                 tan ? mptr(md) : nullptr, want_sums ? mptr(hs) : nullptr, want_sums && htan ? mptr(hds) : nullptr,
                                 stream_of(h)));
        return {m, md, hs, hds};
    }
    auto fn = bf16 ? mdg_cfconv_fwd_bf16 : mdg_cfconv_fwd;
    ok(fn(&net, fptr(d), fptr(dd, "dd"), fptr(h), fptr(hd, "hd"), col.data_ptr<int32_t>(), eid.data_ptr<int32_t>(),
          cnt.data_ptr<int32_t>(), (int)N, (int)col.size(1), mptr(m), tan ? mptr(md) : nullptr, want_sums ? mptr(hs) : nullptr,
          want_sums && htan ? mptr(hds) : nullptr, stream_of(h)));
    return {m, md, hs, hds};
}

// accumulates into d_b / dd_b in place; (gW1, gb1, gW2) are empty unless want_theta
std::tuple<Tensor, Tensor, Tensor> cfconv_bwd(const Tensor& mu, const Tensor& coef, const Tensor& W1, const Tensor& b1,
                                              const Tensor& W2, const Tensor& b2, const Tensor& d, const OptTensor& dd,
                                              const Tensor& nbr, int64_t n_edges, const Tensor& h, const OptTensor& hd,
                                              const OptTensor& mb, const Tensor& mdb, const OptTensor& d_b, Tensor& dd_b,
                                              const OptTensor& n_valid, bool want_theta, bool bf16) {
    const MdgFilterNet net = filter_net(mu, coef, W1, b1, W2, b2);
    const bool r16 = is_bf16(h);                 // bf16 mirrors of the four gathered matrices: mdg_cfconv_bwd_rows16
    check_f32(d, "d"); check_f32(dd_b, "dd_b");
    if (!r16) { check_f32(h, "h"); check_f32(mdb, "mdb"); }
    TORCH_CHECK(nbr.is_cuda() && nbr.scalar_type() == at::kLong && nbr.is_contiguous(), "mdgrad: nbr must be int64 on the device");
    const int64_t G = net.n_gauss, F = net.n_filters;
    const auto fo = h.options().dtype(at::kFloat);
    Tensor gW1 = at::empty({want_theta ? G : 0, G}, fo), gb1 = at::empty({want_theta ? G : 0}, fo);
    Tensor gW2 = at::empty({want_theta ? F : 0, G}, fo);
    Tensor ws = at::empty({want_theta ? std::max<int64_t>(1, mdg_cfconv_bwd_workspace((int)G, (int)F, n_edges)) : 0}, fo);
    const int32_t* nv = nullptr;
    if (n_valid.has_value() && n_valid->defined()) { check_i32(*n_valid, "n_valid"); nv = n_valid->data_ptr<int32_t>(); }
    if (r16) {
        TORCH_CHECK(bf16, "mdgrad: bf16 node rows go with the bf16 filter kernels");
        ok(mdg_cfconv_bwd_rows16(&net, fptr(d), fptr(dd, "dd"), nbr.data_ptr<int64_t>(), n_edges, (int)h.size(0), hptr(h, "h"),
                                 hptr(hd, "hd"), hptr(mb, "mb"), hptr(mdb, "mdb"), const_cast<float*>(fptr(d_b, "d_b")), mptr(dd_b),
                                 want_theta ? mptr(gW1) : nullptr, want_theta ? mptr(gb1) : nullptr, want_theta ? mptr(gW2) : nullptr,
                                 nullptr, nullptr, want_theta ? mptr(ws) : nullptr, nv, stream_of(h)));
        return {gW1, gb1, gW2};
    }
    auto fn = bf16 ? mdg_cfconv_bwd_bf16 : mdg_cfconv_bwd;
    ok(fn(&net, fptr(d), fptr(dd, "dd"), nbr.data_ptr<int64_t>(), n_edges, fptr(h), fptr(hd, "hd"), fptr(mb, "mb"),
          fptr(mdb), const_cast<float*>(fptr(d_b, "d_b")), mptr(dd_b), want_theta ? mptr(gW1) : nullptr, want_theta ? mptr(gb1) : nullptr,
          want_theta ? mptr(gW2) : nullptr, want_theta ? mptr(ws) : nullptr, nv, stream_of(h)));
    return {gW1, gb1, gW2};
}

// out0 = act(x0 B + bias) * mul + res ; out1 = act'(.) (x1 B) + res1 ; (out0, sig0, out1)
std::tuple<Tensor, Tensor, Tensor> dense_ssp(const Tensor& W, bool trans, bool act, const Tensor& x0, const OptTensor& bias,
                                             const OptTensor& mul, const OptTensor& res, const OptTensor& x1,
                                             const OptTensor& res1, bool want_sig) {
    check_f32(W, "W"); check_f32(x0, "x0");
    TORCH_CHECK(W.dim() == 2 && x0.dim() == 2, "mdgrad: dense takes matrices");
    const int64_t N = x0.size(0), K = x0.size(1), M = trans ? W.size(1) : W.size(0);
    TORCH_CHECK((trans ? W.size(0) : W.size(1)) == K, "mdgrad: dense: shape mismatch");
    const bool dual = x1.has_value() && x1->defined();
    Tensor out0 = at::empty({N, M}, x0.options()), sig = at::empty({act && want_sig ? N : 0, M}, x0.options());
    Tensor out1 = at::empty({dual ? N : 0, M}, x0.options());
    ok(mdg_dense(fptr(W), trans, act, (int)N, (int)K, (int)M, fptr(x0), fptr(bias, "bias"), fptr(mul, "mul"), fptr(res, "res"),
                 mptr(out0), act && want_sig ? mptr(sig) : nullptr, fptr(x1, "x1"), fptr(res1, "res1"), dual ? mptr(out1) : nullptr,
                 stream_of(x0)));
    return {out0, sig, out1};
}

std::tuple<Tensor, Tensor> ssp_dual_bwd_t(const Tensor& sa, const Tensor& td, const Tensor& sdb, const Tensor& sb) {
    check_f32(sa, "sa"); check_f32(td, "td"); check_f32(sdb, "sdb"); check_f32(sb, "sb");
    Tensor xdb = at::empty_like(sa), xb = at::empty_like(sa);
    ok(mdg_ssp_dual_bwd_t(fptr(sa), fptr(td), fptr(sdb), fptr(sb), sa.numel(), mptr(xdb), mptr(xb), stream_of(sa)));
    return {xdb, xb};
}

Tensor atb(const Tensor& A, const Tensor& B) {
    check_f32(A, "A"); check_f32(B, "B");
    TORCH_CHECK(A.dim() == 2 && B.dim() == 2 && A.size(0) == B.size(0), "mdgrad: atb takes [E,M] and [E,N]");
    const int64_t E = A.size(0), M = A.size(1), N = B.size(1);
    Tensor Cm = at::empty({M, N}, A.options());
    Tensor ws = at::empty({std::max<int64_t>(1, mdg_atb_workspace(E, (int)M, (int)N))}, A.options());
    ok(mdg_atb(fptr(A), fptr(B), E, (int)M, (int)N, mptr(Cm), mptr(ws), stream_of(A)));
    return Cm;
}

}  // namespace

TORCH_LIBRARY(mdgrad, m) {
    m.def("nbr_build(Tensor pos, float[] cell, float cutoff, Tensor? mask, int max_nbr, int group, bool cell_list) -> "
          "(Tensor, Tensor, Tensor, Tensor)");
    m.def("pair_force(Tensor pos, float[] cell, Tensor col, Tensor shift, Tensor cnt, int[] term_i, float[] term_f, Tensor? mask, "
          "Tensor? theta) -> (Tensor, Tensor, Tensor)");
    m.def("pair_hvp(Tensor pos, float[] cell, Tensor col, Tensor shift, Tensor cnt, int[] term_i, float[] term_f, Tensor? mask, "
          "Tensor? theta, Tensor w) -> (Tensor, Tensor)");
    m.def("nhc_vv_forward(Tensor v0, Tensor q0, Tensor? pv0, Tensor mass, Tensor t, Tensor? theta, int[] iprm, float[] fprm, "
          "float[] cell, int[] terms_i, float[] terms_f, int n_theta_total) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("nhc_vv_adjoint(Tensor v_t, Tensor q_t, Tensor? pv_t, Tensor? g_v, Tensor? g_q, Tensor? g_pv, Tensor mass, Tensor t, "
          "Tensor? theta, int[] iprm, float[] fprm, float[] cell, int[] terms_i, float[] terms_f, int n_theta_total) -> "
          "(Tensor, Tensor, Tensor, Tensor)");
    m.def("rdf_fwd(Tensor xyz, float[] cell, float cutoff, Tensor? mask, Tensor mu, float spacing, float coeff) -> Tensor");
    m.def("rdf_bwd(Tensor xyz, float[] cell, float cutoff, Tensor? mask, Tensor mu, float spacing, float coeff, Tensor g_raw) -> Tensor");
    m.def("edge_geom(Tensor x, Tensor? w, Tensor nbr, Tensor offsets) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("edge_geom_bwd(Tensor? d_b, Tensor dd_b, Tensor? d, Tensor? dd, Tensor uhat, Tensor? ddel, Tensor col, Tensor eid, "
          "Tensor cnt) -> (Tensor, Tensor)");
    m.def("cfconv_fwd(Tensor mu, Tensor coef, Tensor W1, Tensor b1, Tensor W2, Tensor b2, bool bf16, Tensor d, Tensor? dd, Tensor h, "
          "Tensor? hd, Tensor col, Tensor eid, Tensor cnt, bool want_sums) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("cfconv_bwd(Tensor mu, Tensor coef, Tensor W1, Tensor b1, Tensor W2, Tensor b2, Tensor d, Tensor? dd, Tensor nbr, "
          "int n_edges, Tensor h, Tensor? hd, Tensor? mb, Tensor mdb, Tensor(a!)? d_b, Tensor(b!) dd_b, Tensor? n_valid, "
          "bool want_theta, bool bf16=False) -> (Tensor, Tensor, Tensor)");
    m.def("dense_ssp(Tensor W, bool trans, bool act, Tensor x0, Tensor? bias, Tensor? mul, Tensor? res, Tensor? x1, Tensor? res1, "
          "bool want_sig) -> (Tensor, Tensor, Tensor)");
    m.def("ssp_dual_bwd_t(Tensor sa, Tensor td, Tensor sdb, Tensor sb) -> (Tensor, Tensor)");
    m.def("atb(Tensor A, Tensor B) -> Tensor");
}

TORCH_LIBRARY_IMPL(mdgrad, CUDA, m) {      // (the HIP backend registers under the CUDA dispatch key in PyTorch-ROCm)
    m.impl("nbr_build", nbr_build);
    m.impl("pair_force", pair_force);
    m.impl("pair_hvp", pair_hvp);
    m.impl("nhc_vv_forward", nhc_vv_forward);
    m.impl("nhc_vv_adjoint", nhc_vv_adjoint);
    m.impl("rdf_fwd", rdf_fwd);
    m.impl("rdf_bwd", rdf_bwd);
    m.impl("edge_geom", edge_geom);
    m.impl("edge_geom_bwd", edge_geom_bwd);
    m.impl("cfconv_fwd", cfconv_fwd);
    m.impl("cfconv_bwd", cfconv_bwd);
    m.impl("dense_ssp", dense_ssp);
    m.impl("ssp_dual_bwd_t", ssp_dual_bwd_t);
    m.impl("atb", atb);
}
