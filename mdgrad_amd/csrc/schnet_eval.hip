// One SchNet force / force-vjp evaluation enqueued from C++ (round 6).
//
// mdgrad_amd/nn/analytic.py derives the force F = -dU/dx and, for an atom vector w, d(w.F)/dx and d(w.F)/dtheta of the
// SchNet energy (nff/nn/models/schnet.py:113-171 under torchmd/interface.py:86-136) by hand -- primal (+ tangent) sweep, turn
// at the readout, reverse sweep -- on top of the kernels of this library: the fused interaction block (cfconv_fused.hip), the
// row chains of the node-level layers (rowchain.hip), the batched parameter-gradient reductions (gradjobs.hip).  Until
// round 5 that sequence -- 20 to 30 launches per evaluation, three evaluations per MD step (torchmd/sovlers.py:106-168,
// 211-293) -- was issued launch by launch from Python (ctypes marshalling, fresh torch tensors for every intermediate): the
// stacked 8 x 4 096-bead pass was host-coupled, the launch thread finishing 0.05 ms ahead of the GPU.  Here the whole
// evaluation is ONE C-ABI call: the caller describes the network and the topology once (MdgSchnetPlan, device pointers) and
// hands over one workspace; the launches, their order and their arguments are exactly those of
// analytic._force_chain / _force_vjp_chain (same kernels, same operands: the results are bitwise the Python-driven ones,
// tests/test_gpu_schnet_plan.py), issued back to back from this loop.
#include "common.hpp"
#include <vector>

namespace {

// (the per-edge accumulators are cleared by the geometry kernel, not by hipMemsetAsync: the evaluation is captured into HIP
//  graphs, and a memset NODE between kernel nodes was seen to race with its neighbours on ROCm 7.2 -- flaky non-finite replays)
struct Arena {
    float* base;
    size_t off;
    float* take(size_t n) {
        float* p = base ? base + off : nullptr;
        off += (n + 63) / 64 * 64;                          // 256-byte aligned pieces
        return p;
    }
    uint16_t* take16(size_t n) { return reinterpret_cast<uint16_t*>(take((n + 1) / 2)); }
};

// Every intermediate of one evaluation, carved from the workspace in a fixed order (the same walk sizes it).
struct Bufs {
    float *d, *uhat, *dd, *ddel, *both;                     // per edge; both = [2][E]: d_b, dd_b (first order: dd_b only)
    struct Layer {
        float *m, *md, *hsum, *hdsum, *t, *su, *td, *r, *rd;  // forward (r, rd: the block's OUTPUT rows)
        float *hn, *hnd; uint16_t *hn16, *hnd16;            // the NEXT block's filtered rows (and / or their bf16 mirrors)
        float *hdb, *hb;                                    // reverse: adjoints of (hd, h)
        float *g0, *g1, *e0, *e1, *f0, *f1; uint16_t *f016, *f116;   // reverse chain below this block (idx > 0) / rb of block 0 in g0
        float *gW1, *gb1, *gW2, *gb2;
        uint16_t *st_s, *st_sd;                             // the block's filter stash of this evaluation (null: recompute)
    } L[MDG_SCHNET_MAX_LAYERS];
    float *y0, *y1, *ysig, *ypre0, *ypre1, *g0, *g1, *e0, *e1, *f0, *f1; uint16_t *f016, *f116;   // the turn
    float *cf_ws, *en_ws, *gj_ws;
    size_t gj_cap;                                          // floats left for the grad-job workspace
};

size_t carve(const MdgSchnetPlan& P, bool dual, bool theta, float* base, Bufs& B) {   // -> floats used before the grad-job workspace
    Arena a{base, 0};
    const size_t N = (size_t)P.n_atoms, E = (size_t)P.n_edges, A = (size_t)P.n_atom_basis, H = (size_t)P.n_readout;
    B = Bufs{};
    B.d = a.take(E); B.uhat = a.take(3 * E);
    if (dual) { B.dd = a.take(E); B.ddel = a.take(3 * E); }
    B.both = a.take(2 * E);
    long long cf = 1;
    for (int i = 0; i < P.n_layers; ++i) {
        const MdgSchnetLayer& S = P.layer[i];
        const size_t F = (size_t)S.filt.n_filters, G = (size_t)S.filt.n_gauss;
        Bufs::Layer& L = B.L[i];
        const bool sums = theta && !S.b2col;
        // (measured, profiles/r06_cfconv_kbench_stash.txt: with bf16 node rows the stashed sweeps take 38 / 48 / 55 us against
        //  53 / 76 / 88 recomputing; with f32 rows the sweeps are gather-bound either way and the producer is a net loss --
        //  so the stash goes with rows16.  The neighbour sums ride on the recomputing kernels.)
        if (P.stash && S.bf16 && S.rows16 && !sums) {
            const size_t W = (size_t)mdg_cfconv_stash_width((int)G);
            L.st_s = a.take16(E * W);
            if (dual) L.st_sd = a.take16(E * W);
        }
        L.m = a.take(N * F);
        if (dual) L.md = a.take(N * F);
        if (sums) { L.hsum = a.take(N * F); if (dual && i > 0) L.hdsum = a.take(N * F); }
        L.t = a.take(N * A); L.su = a.take(N * A);
        if (dual) L.td = a.take(N * A);
        L.r = a.take(N * A);
        if (dual) L.rd = a.take(N * A);
        if (i + 1 < P.n_layers) {
            const MdgSchnetLayer& Sn = P.layer[i + 1];
            const size_t Fn = (size_t)Sn.filt.n_filters;
            if (Sn.rows16) { L.hn16 = a.take16(N * Fn); if (dual) L.hnd16 = a.take16(N * Fn); }
            else { L.hn = a.take(N * Fn); if (dual) L.hnd = a.take(N * Fn); }
        }
        // reverse
        L.hb = a.take(N * F);
        if (dual) L.hdb = a.take(N * F);
        if (i > 0) {
            const MdgSchnetLayer& Sp = P.layer[i - 1];
            const size_t Fp = (size_t)Sp.filt.n_filters;
            L.g0 = a.take(N * A); L.f0 = a.take(N * Fp);
            if (Sp.rows16) L.f016 = a.take16(N * Fp);
            if (dual) {
                L.g1 = a.take(N * A); L.e0 = a.take(N * A); L.e1 = a.take(N * A); L.f1 = a.take(N * Fp);
                if (Sp.rows16) L.f116 = a.take16(N * Fp);
            }
        } else {
            L.g0 = a.take(N * A);
        }
        if (theta) {
            L.gW1 = a.take(G * G); L.gb1 = a.take(G); L.gW2 = a.take(F * G); L.gb2 = a.take(F);
            const long long w = mdg_cfconv_bwd_workspace((int)G, (int)F, (int64_t)E);
            if (w > cf) cf = w;
        }
    }
    const MdgSchnetLayer& SL = P.layer[P.n_layers - 1];
    const size_t FL = (size_t)SL.filt.n_filters;
    B.y0 = a.take(N * H); B.ysig = a.take(N * H); B.ypre0 = a.take(N * H);
    if (dual) { B.y1 = a.take(N * H); B.ypre1 = a.take(N * H); }
    B.g0 = a.take(N * A); B.e0 = a.take(N * A); B.f0 = a.take(N * FL);
    if (SL.rows16) B.f016 = a.take16(N * FL);
    if (dual) {
        B.g1 = a.take(N * A); B.e1 = a.take(N * A); B.f1 = a.take(N * FL);
        if (SL.rows16) B.f116 = a.take16(N * FL);
    }
    B.cf_ws = a.take((size_t)cf);
    {   // the energy's column-sum job (first-order passes ask for it too)
        MdgGradJob j{};
        j.A = reinterpret_cast<const float*>(256); j.rows = (int64_t)N; j.m = (int32_t)H; j.kind = MDG_GRAD_COLSUM;
        const long long w = mdg_grad_jobs_workspace(&j, 1);
        B.en_ws = a.take((size_t)(w > 0 ? w : 1));
    }
    B.gj_ws = a.take(0);                                     // the parameter-gradient reductions take what follows (sized by run_vjp's dry pass)
    B.gj_cap = 0;
    return a.off;
}

struct Chain {
    MdgChainStage s[MDG_CHAIN_MAX_STAGES];
    int n = 0;
    MdgChainStage& add(const float* W, int K, int M, int trans, int act, int mode) {
        MdgChainStage& x = s[n++];
        x = MdgChainStage{};
        x.W = W; x.K = K; x.M = M; x.trans = trans; x.act = act; x.mode = mode;
        return x;
    }
};

struct Jobs {
    std::vector<MdgGradJob> j;
    void add(int kind, int64_t off, int64_t rows, int m, int n, const float* A, const float* Bm, const float* A2, const float* B2,
             const int64_t* row_map = nullptr) {
        MdgGradJob x{};
        x.A = A; x.B = Bm; x.A2 = A2; x.B2 = B2; x.row_map = row_map; x.rows = rows; x.m = m; x.n = n; x.kind = kind; x.out_off = off;
        j.push_back(x);
    }
    void atb(int64_t off, int64_t rows, int m, int n, const float* A, const float* Bm, const float* A2 = nullptr, const float* B2 = nullptr,
             const int64_t* row_map = nullptr) { add(MDG_GRAD_ATB, off, rows, m, n, A, Bm, A2, B2, row_map); }
    void colsum(int64_t off, int64_t rows, int m, const float* A, const float* Bm = nullptr, const float* A2 = nullptr, const float* B2 = nullptr) {
        add(MDG_GRAD_COLSUM, off, rows, m, 0, A, Bm, A2, B2);
    }
    void axpy(int64_t off, int m, const float* A) { add(MDG_GRAD_AXPY, off, 1, m, 0, A, nullptr, nullptr, nullptr); }
};

int check_plan(const MdgSchnetPlan* P) {
    MDG_CHECK_ARG(P && P->n_layers >= 1 && P->n_layers <= MDG_SCHNET_MAX_LAYERS, "schnet plan: 1..%d interaction blocks", MDG_SCHNET_MAX_LAYERS);
    MDG_CHECK_ARG(P->n_atoms > 0 && P->n_edges >= 0 && P->n_atom_basis > 0 && P->n_readout > 0, "schnet plan: bad sizes");
    MDG_CHECK_ARG(P->n_atom_basis <= MDG_CHAIN_MAX_WIDTH && P->n_readout <= MDG_CHAIN_MAX_WIDTH, "schnet plan: node widths up to %d", MDG_CHAIN_MAX_WIDTH);
    MDG_CHECK_ARG(P->L1 && P->L2 && P->r0 && P->h0 && P->nbr && P->col && P->eid && P->cnt && P->ws, "schnet plan: null pointer");
    for (int i = 0; i < P->n_layers; ++i) {
        const MdgSchnetLayer& S = P->layer[i];
        MDG_CHECK_ARG(S.Wn && S.U1 && S.U2 && S.filt.mu && S.filt.W1 && S.filt.W2, "schnet plan: null weights in block %d", i);
        MDG_CHECK_ARG(mdg_cfconv_supported(S.filt.n_gauss, S.filt.n_filters), "schnet plan: block %d is outside the fused kernels", i);
        MDG_CHECK_ARG(S.filt.n_filters <= MDG_CHAIN_MAX_WIDTH, "schnet plan: filter width");
        MDG_CHECK_ARG(!S.rows16 || (S.bf16 && S.bf16_rev && mdg_cfconv_rows16_supported(S.filt.n_gauss, S.filt.n_filters)), "schnet plan: rows16 needs the bf16 kernels");
        MDG_CHECK_ARG(!S.b2col || mdg_cfconv_bias_column(S.filt.n_gauss), "schnet plan: no spare filter column for the bias gradient");
    }
    MDG_CHECK_ARG(!P->layer[0].rows16 || P->h0_16, "schnet plan: the first block gathers a bf16 mirror of h0");
    return MDG_OK;
}

#define MDG_TRY(call) do { const int rc_ = (call); if (rc_ != MDG_OK) return rc_; } while (0)
// (dry pass: the same walk without launches -- it yields the parameter-gradient job list, whose workspace the planner of
//  gradjobs.hip sizes from the shapes alone)
#define MDG_RUN(call) do { if (!dry) MDG_TRY(call); } while (0)

// geometry of the half list + the zero fill of the per-edge accumulators, one launch
int geom(const MdgSchnetPlan& P, const Bufs& B, const float* x, const float* w, size_t zero_n, void* st) {
    return mdg_edge_geom_prepare(x, w, P.nbr, P.offsets, P.n_edges, P.masked ? &P.cell : nullptr, P.cutoff, B.d, B.uhat,
                                 w ? B.dd : nullptr, w ? B.ddel : nullptr, B.both, (int64_t)zero_n, st);
}

int conv_fwd(const MdgSchnetPlan& P, const MdgSchnetLayer& S, const float* d, const float* dd, const void* h, const void* hd,
             float* m, float* md, float* hsum, float* hdsum, void* st, const Bufs::Layer* stl = nullptr) {
    if (stl && stl->st_s && !hsum)                              // second filter layer's operands from this evaluation's stash
        return mdg_cfconv_fwd_stashed(&S.filt, stl->st_s, dd ? stl->st_sd : nullptr, d, h, hd, P.col, P.eid, P.cnt, P.n_atoms, P.max_nbr,
                                      m, md, S.rows16, st);
    if (S.rows16)
        return mdg_cfconv_fwd_rows16(&S.filt, d, dd, (const uint16_t*)h, (const uint16_t*)hd, P.col, P.eid, P.cnt, P.n_atoms, P.max_nbr,
                                     m, md, hsum, hdsum, st);
    return (S.bf16 ? mdg_cfconv_fwd_bf16 : mdg_cfconv_fwd)(&S.filt, d, dd, (const float*)h, (const float*)hd, P.col, P.eid, P.cnt,
                                                           P.n_atoms, P.max_nbr, m, md, hsum, hdsum, st);
}

int conv_bwd(const MdgSchnetPlan& P, const MdgSchnetLayer& S, const Bufs& B, const float* dd, const void* h, const void* hd, const void* mb,
             const void* mdb, float* d_b, float* dd_b, const Bufs::Layer* th, void* st) {
    if (th && S.b2col) {
        const int flags = (S.bf16_rev ? MDG_CFCONV_BF16 : 0) | (S.rows16 ? MDG_CFCONV_ROWS16 : 0);
        return mdg_cfconv_bwd_theta(&S.filt, B.d, dd, P.nbr, P.n_edges, P.n_atoms, h, hd, mb, mdb, d_b, dd_b, th->gW1, th->gb1, th->gW2,
                                    th->gb2, nullptr, nullptr, B.cf_ws, P.n_valid, flags, st);
    }
    float *gW1 = th ? th->gW1 : nullptr, *gb1 = th ? th->gb1 : nullptr, *gW2 = th ? th->gW2 : nullptr;
    if (S.rows16)
        return mdg_cfconv_bwd_rows16(&S.filt, B.d, dd, P.nbr, P.n_edges, P.n_atoms, (const uint16_t*)h, (const uint16_t*)hd,
                                     (const uint16_t*)mb, (const uint16_t*)mdb, d_b, dd_b, gW1, gb1, gW2, nullptr, nullptr, th ? B.cf_ws : nullptr,
                                     P.n_valid, st);
    return (S.bf16_rev ? mdg_cfconv_bwd_bf16 : mdg_cfconv_bwd)(&S.filt, B.d, dd, P.nbr, P.n_edges, (const float*)h, (const float*)hd,
                                                           (const float*)mb, (const float*)mdb, d_b, dd_b, gW1, gb1, gW2,
                                                           th ? B.cf_ws : nullptr, P.n_valid, st);
}

// Primal (+ tangent along w) sweep and the turn at the readout: analytic._chain_forward.
int forward_and_turn(const MdgSchnetPlan& P, Bufs& B, const float* x, const float* w, bool dual, bool want_pre0, void* st, bool dry,
                     size_t zero_n) {                            // zero_n: floats of B.both the reverse sweeps accumulate into
    const int N = P.n_atoms, A = P.n_atom_basis, H = P.n_readout;
    // (tried and dropped, round 6: geometry + accumulator fill + every block's stash in ONE launch -- 3 599 MD steps/s on the
    //  stacked pass against 3 657 with the separate launches: one wave of four doing the latency-bound geometry of a 64-edge
    //  tile serialises what edge_geom_kernel does with the whole chip)
    MDG_RUN(geom(P, B, x, dual ? w : nullptr, zero_n, st));
    const float *r = P.r0, *rd = nullptr;
    const void* hg = P.layer[0].rows16 ? (const void*)P.h0_16 : (const void*)P.h0;
    const void* hgd = nullptr;
    for (int i = 0; i < P.n_layers; ++i) {
        const MdgSchnetLayer& S = P.layer[i];
        Bufs::Layer& L = B.L[i];
        const int F = S.filt.n_filters;
        if (L.st_s)
            MDG_RUN(mdg_cfconv_filter_stash(&S.filt, B.d, dual ? B.dd : nullptr, P.n_edges, P.n_valid, L.st_s, dual ? L.st_sd : nullptr, st));
        MDG_RUN(conv_fwd(P, S, B.d, dual ? B.dd : nullptr, hg, hgd, L.m, dual ? L.md : nullptr, L.hsum, hgd ? L.hdsum : nullptr, st, &L));
        Chain c;
        {   // update MLP: t = ssp(U1 m + c1), su = sigmoid(.), td = su (U1 md)
            MdgChainStage& s = c.add(S.U1, F, A, 0, 1, MDG_CHAIN_NONE);
            s.bias = S.c1; s.in0 = L.m; s.in1 = dual ? L.md : nullptr; s.out0 = L.t; s.out1 = dual ? L.td : nullptr; s.sig = L.su;
        }
        {   // residual (schnet.py:149-151)
            MdgChainStage& s = c.add(S.U2, A, A, 0, 0, MDG_CHAIN_NONE);
            s.bias = S.c2; s.res0 = r; s.res1 = rd; s.out0 = L.r; s.out1 = dual ? L.rd : nullptr;
        }
        if (i + 1 < P.n_layers) {
            const MdgSchnetLayer& Sn = P.layer[i + 1];
            MdgChainStage& s = c.add(Sn.Wn, A, Sn.filt.n_filters, 0, 0, MDG_CHAIN_NONE);
            s.bias = Sn.bn; s.out0 = L.hn; s.out1 = dual ? L.hnd : nullptr; s.out0_h = L.hn16; s.out1_h = dual ? L.hnd16 : nullptr;
        } else {
            {   // readout layer + head (MDG_CHAIN_HEAD): ydb = sy L2, yb = (1 - sy) syd L2
                MdgChainStage& s = c.add(P.L1, A, H, 0, 1, MDG_CHAIN_HEAD);
                s.bias = P.l1; s.aux0 = P.L2; s.out0 = B.y0; s.out1 = dual ? B.y1 : nullptr; s.sig = B.ysig;
                s.pre0 = want_pre0 ? B.ypre0 : nullptr; s.pre1 = dual ? B.ypre1 : nullptr;
            }
            {   // readout^T: rdb, rb
                MdgChainStage& s = c.add(P.L1, H, A, 1, 0, MDG_CHAIN_NONE);
                s.out0 = B.g0; s.out1 = dual ? B.g1 : nullptr;
            }
            {   // U2^T through the update MLP's activation: udb, ub
                MdgChainStage& s = c.add(S.U2, A, A, 1, 0, dual ? MDG_CHAIN_SSP_BWD : MDG_CHAIN_MUL);
                s.aux0 = L.su; s.aux1 = dual ? L.td : nullptr; s.out0 = B.e0; s.out1 = dual ? B.e1 : nullptr;
            }
            {   // U1^T: mdb, mb (+ the mirrors the rows16 kernels gather)
                MdgChainStage& s = c.add(S.U1, A, F, 1, 0, MDG_CHAIN_NONE);
                s.out0 = B.f0; s.out1 = dual ? B.f1 : nullptr; s.out0_h = B.f016; s.out1_h = dual ? B.f116 : nullptr;
            }
        }
        MDG_RUN(mdg_row_chain(c.s, c.n, N, (dual ? MDG_CHAIN_DUAL : 0) | (P.chain_x3 & (MDG_CHAIN_X3 | MDG_CHAIN_X6)), st));
        r = L.r; rd = dual ? L.rd : nullptr;
        if (i + 1 < P.n_layers) {
            const bool r16 = P.layer[i + 1].rows16 != 0;
            hg = r16 ? (const void*)L.hn16 : (const void*)L.hn;
            hgd = dual ? (r16 ? (const void*)L.hnd16 : (const void*)L.hnd) : nullptr;
        }
    }
    return MDG_OK;
}

// what block i's kernels gather as (h, hd): the first block's persistent rows or the previous block's chain outputs
void gathered_rows(const MdgSchnetPlan& P, const Bufs& B, int i, bool dual, const void*& h, const void*& hd) {
    const bool r16 = P.layer[i].rows16 != 0;
    if (i == 0) { h = r16 ? (const void*)P.h0_16 : (const void*)P.h0; hd = nullptr; return; }
    const Bufs::Layer& Lp = B.L[i - 1];
    h = r16 ? (const void*)Lp.hn16 : (const void*)Lp.hn;
    hd = dual ? (r16 ? (const void*)Lp.hnd16 : (const void*)Lp.hnd) : nullptr;
}

// analytic._force_vjp_chain (dry: no launches; -> *gj_need = floats the parameter-gradient reductions ask for)
int run_vjp(const MdgSchnetPlan& P, Bufs& B, const float* x, const float* w, float* force, float* dwf, float* theta_flat, bool theta,
            float alpha, const float* t, const int64_t* idx_dev, float* energy_colsum, void* stream, bool dry, long long* gj_need) {
    hipStream_t st = (hipStream_t)stream;
    MDG_TRY(forward_and_turn(P, B, x, w, true, energy_colsum != nullptr, stream, dry, 2 * (size_t)P.n_edges));
    const int N = P.n_atoms, A = P.n_atom_basis, H = P.n_readout;
    const int nl = P.n_layers;
    Jobs J;
    const float *ydb = B.y0, *yb = B.y1;
    const float *rdb = B.g0, *rb = B.g1, *udb = B.e0, *ub = B.e1, *mdb = B.f0, *mb = B.f1;
    const void *mdg = P.layer[nl - 1].rows16 ? (const void*)B.f016 : (const void*)B.f0;
    const void *mg = P.layer[nl - 1].rows16 ? (const void*)B.f116 : (const void*)B.f1;
    if (theta) {
        const float* r_fin = B.L[nl - 1].r;
        const float* rd_fin = B.L[nl - 1].rd;
        J.colsum(P.off_L2, N, H, B.ypre1);                                           // (readout's last bias: U_dot does not see it)
        J.atb(P.off_L1, N, H, A, yb, r_fin, ydb, rd_fin);
        J.colsum(P.off_l1, N, H, yb);
    }
    float *d_b = B.both, *dd_b = B.both + P.n_edges;
    for (int idx = nl - 1; idx >= 0; --idx) {
        const MdgSchnetLayer& S = P.layer[idx];
        Bufs::Layer& L = B.L[idx];
        const int F = S.filt.n_filters, G = S.filt.n_gauss;
        const float* r_in = idx == 0 ? P.r0 : B.L[idx - 1].r;
        const float* rd_in = idx == 0 ? nullptr : B.L[idx - 1].rd;
        if (theta) {
            J.atb(S.off_U2, N, A, A, rb, L.t, rdb, L.td);
            J.colsum(S.off_c2, N, A, rb);
            J.atb(S.off_U1, N, A, F, udb, L.md, ub, L.m);
            J.colsum(S.off_c1, N, A, ub);
        }
        const void *h, *hd;
        gathered_rows(P, B, idx, true, h, hd);
        MDG_RUN(conv_bwd(P, S, B, B.dd, h, hd, mg, mdg, d_b, dd_b, theta ? &L : nullptr, stream));
        if (theta) {
            J.axpy(S.off_W1, G * G, L.gW1);
            J.axpy(S.off_b1, G, L.gb1);
            J.axpy(S.off_W2, F * G, L.gW2);
            if (S.b2col) J.axpy(S.off_b2, F, L.gb2);
            else if (L.hdsum) J.colsum(S.off_b2, N, F, mb, L.hsum, mdb, L.hdsum);    // sum_e W_b[e] = sum_n mb_n (.) sum_j h_j (+ tangent half)
            else J.colsum(S.off_b2, N, F, mb, L.hsum);
        }
        if (theta || idx > 0) {
            // the aggregation is symmetric in the adjacency: fed (mdb, mb) the forward kernel returns the adjoints (hdb, hb) of (hd, h)
            MDG_RUN(conv_fwd(P, S, B.d, B.dd, mdg, mg, L.hdb, L.hb, nullptr, nullptr, stream, &L));
            if (theta) {
                if (rd_in) J.atb(S.off_Wn, N, F, A, L.hb, r_in, L.hdb, rd_in);
                else J.atb(S.off_Wn, N, F, A, L.hb, r_in);
                J.colsum(S.off_bn, N, F, L.hb);
            }
            if (idx > 0) {
                const MdgSchnetLayer& Sp = P.layer[idx - 1];
                const Bufs::Layer& Lp = B.L[idx - 1];
                Chain c;
                { MdgChainStage& s = c.add(S.Wn, F, A, 1, 0, MDG_CHAIN_NONE); s.in0 = L.hdb; s.in1 = L.hb; s.res0 = rdb; s.res1 = rb; s.out0 = L.g0; s.out1 = L.g1; }
                { MdgChainStage& s = c.add(Sp.U2, A, A, 1, 0, MDG_CHAIN_SSP_BWD); s.aux0 = Lp.su; s.aux1 = Lp.td; s.out0 = L.e0; s.out1 = L.e1; }
                { MdgChainStage& s = c.add(Sp.U1, A, Sp.filt.n_filters, 1, 0, MDG_CHAIN_NONE); s.out0 = L.f0; s.out1 = L.f1; s.out0_h = L.f016; s.out1_h = L.f116; }
                MDG_RUN(mdg_row_chain(c.s, c.n, N, MDG_CHAIN_DUAL | (P.chain_x3 & (MDG_CHAIN_X3 | MDG_CHAIN_X6)), stream));
                rdb = L.g0; rb = L.g1; udb = L.e0; ub = L.e1; mdb = L.f0; mb = L.f1;
                mdg = Sp.rows16 ? (const void*)L.f016 : (const void*)L.f0;
                mg = Sp.rows16 ? (const void*)L.f116 : (const void*)L.f1;
            } else {
                // below the first block only the embedding rows' adjoint in U_dot is left (r_dot^0 = 0)
                Chain c;
                { MdgChainStage& s = c.add(S.Wn, F, A, 1, 0, MDG_CHAIN_NONE); s.in0 = L.hb; s.res0 = rb; s.out0 = L.g0; }
                MDG_RUN(mdg_row_chain(c.s, c.n, N, P.chain_x3 & (MDG_CHAIN_X3 | MDG_CHAIN_X6), stream));
                rb = L.g0;
            }
        }
    }
    // dd_b = dU/dd: force and d(w.F)/dx from one scatter
    MDG_RUN(mdg_edge_geom_bwd(d_b, dd_b, B.d, B.dd, B.uhat, B.ddel, P.col, P.eid, P.cnt, N, P.max_nbr, force, dwf, stream));
    long long need = 0;
    if (theta) {
        // embedding rows: one-hot(z)^T rb, row s of the product -> row uniq[s] of the table
        J.atb(P.off_embed, N, P.n_species, A, P.onehot, rb, nullptr, nullptr, P.uniq);
        for (size_t c0 = 0; c0 < J.j.size(); c0 += MDG_GRAD_JOBS_MAX) {
            const int n = (int)((J.j.size() - c0) < (size_t)MDG_GRAD_JOBS_MAX ? (J.j.size() - c0) : (size_t)MDG_GRAD_JOBS_MAX);
            const long long wsn = mdg_grad_jobs_workspace(J.j.data() + c0, n);
            if (wsn < 0) return MDG_EINVAL;
            if (wsn > need) need = wsn;
            if (!dry) {
                MDG_CHECK_ARG((size_t)wsn <= B.gj_cap, "schnet_force_vjp: %zu floats left for the parameter-gradient reductions, %lld needed",
                              B.gj_cap, wsn);
                MDG_TRY(mdg_grad_jobs(J.j.data() + c0, n, theta_flat, alpha, t, idx_dev, 1, B.gj_ws, stream));
            }
        }
    }
    if (gj_need) *gj_need = need;
    if (energy_colsum && !dry) {
        MdgGradJob j{};
        j.A = B.ypre0; j.rows = N; j.m = H; j.kind = MDG_GRAD_COLSUM; j.out_off = 0;
        MDG_TRY(mdg_grad_jobs(&j, 1, energy_colsum, 1.0f, nullptr, nullptr, 0, B.en_ws, stream));
    }
    return MDG_OK;
}

}  // namespace

extern "C" int64_t mdg_schnet_plan_sizeof(void) { return (int64_t)sizeof(MdgSchnetPlan); }

// floats of workspace an evaluation needs: dual = 0 mdg_schnet_force, dual = 1 mdg_schnet_force_vjp (theta: with parameter gradients)
extern "C" int64_t mdg_schnet_workspace(const MdgSchnetPlan* plan, int dual, int theta) {
    if (!plan || plan->n_layers < 1 || plan->n_layers > MDG_SCHNET_MAX_LAYERS || plan->n_atoms <= 0 || plan->n_edges < 0) return -1;
    Bufs B;
    float* fake = reinterpret_cast<float*>(4096);               // (never dereferenced: the walk only hands pointers on)
    size_t used = carve(*plan, dual != 0, dual != 0 && theta != 0, fake, B);
    long long gj = 0;
    if (dual && theta) {
        if (run_vjp(*plan, B, fake, fake, fake, fake, fake, true, 1.f, nullptr, nullptr, nullptr, nullptr, true, &gj) != MDG_OK) return -1;
    }
    return (int64_t)(used + (size_t)((gj + 63) / 64 * 64) + 64);
}

// analytic._force_chain: F = -dU/dx (energy_colsum nullable: the column sums of the readout's activations, from which the
// caller gets U = L2 . colsum + N l2)
extern "C" int mdg_schnet_force(const MdgSchnetPlan* plan, const float* x, float* force, float* energy_colsum, void* stream) {
    MDG_TRY(check_plan(plan));
    MDG_CHECK_ARG(x && force, "schnet_force: null buffer");
    const MdgSchnetPlan& P = *plan;
    const bool dry = false;
    Bufs B;
    const size_t need = carve(P, false, false, P.ws, B);
    MDG_CHECK_ARG((int64_t)need <= P.ws_floats, "schnet_force: workspace of %lld floats, %zu needed", (long long)P.ws_floats, need);
    hipStream_t st = (hipStream_t)stream;
    MDG_TRY(forward_and_turn(P, B, x, nullptr, false, energy_colsum != nullptr, stream, dry, (size_t)P.n_edges));
    const int N = P.n_atoms, A = P.n_atom_basis;
    float* dU_dd = B.both;
    const float* rb = B.g0;
    const void* mg = P.layer[P.n_layers - 1].rows16 ? (const void*)B.f016 : (const void*)B.f0;
    for (int idx = P.n_layers - 1; idx >= 0; --idx) {
        const MdgSchnetLayer& S = P.layer[idx];
        Bufs::Layer& L = B.L[idx];
        const void *h, *hd;
        gathered_rows(P, B, idx, false, h, hd);
        MDG_TRY(conv_bwd(P, S, B, nullptr, h, nullptr, nullptr, mg, nullptr, dU_dd, nullptr, stream));
        if (idx > 0) {                                           // (the embedding below block 0 is not needed)
            MDG_TRY(conv_fwd(P, S, B.d, nullptr, mg, nullptr, L.hb, nullptr, nullptr, nullptr, stream, &L));
            const MdgSchnetLayer& Sp = P.layer[idx - 1];
            const Bufs::Layer& Lp = B.L[idx - 1];
            Chain c;
            { MdgChainStage& s = c.add(S.Wn, S.filt.n_filters, A, 1, 0, MDG_CHAIN_NONE); s.in0 = L.hb; s.res0 = rb; s.out0 = L.g0; }
            { MdgChainStage& s = c.add(Sp.U2, A, A, 1, 0, MDG_CHAIN_MUL); s.aux0 = Lp.su; }
            { MdgChainStage& s = c.add(Sp.U1, A, Sp.filt.n_filters, 1, 0, MDG_CHAIN_NONE); s.out0 = L.f0; s.out0_h = L.f016; }
            MDG_TRY(mdg_row_chain(c.s, c.n, N, P.chain_x3 & (MDG_CHAIN_X3 | MDG_CHAIN_X6), stream));
            rb = L.g0;
            mg = Sp.rows16 ? (const void*)L.f016 : (const void*)L.f0;
        }
    }
    MDG_TRY(mdg_edge_geom_bwd(nullptr, dU_dd, nullptr, nullptr, B.uhat, nullptr, P.col, P.eid, P.cnt, N, P.max_nbr, force, nullptr, stream));
    if (energy_colsum) {
        MdgGradJob j{};
        j.A = B.ypre0; j.rows = N; j.m = P.n_readout; j.kind = MDG_GRAD_COLSUM; j.out_off = 0;
        MDG_TRY(mdg_grad_jobs(&j, 1, energy_colsum, 1.0f, nullptr, nullptr, 0, B.en_ws, stream));
    }
    return MDG_OK;
}

// analytic._force_vjp_chain: F, d(w.F)/dx and -- theta_flat != NULL -- d(w.F)/dtheta accumulated into the flat parameter-gradient
// buffer with weight alpha * (t ? t[*idx] - t[*idx - 1] : 1)  (ops.ThetaAccum; callers pass alpha = -1: w.F = -U_dot)
extern "C" int mdg_schnet_force_vjp(const MdgSchnetPlan* plan, const float* x, const float* w, float* force, float* dwf,
                                    float* theta_flat, float alpha, const float* t, const int64_t* idx_dev, float* energy_colsum,
                                    void* stream) {
    MDG_TRY(check_plan(plan));
    MDG_CHECK_ARG(x && w && force && dwf, "schnet_force_vjp: null buffer");
    MDG_CHECK_ARG((t == nullptr) == (idx_dev == nullptr), "schnet_force_vjp: the time grid and the frame index go together");
    const MdgSchnetPlan& P = *plan;
    const bool theta = theta_flat != nullptr;
    MDG_CHECK_ARG(!theta || (P.onehot && P.uniq && P.n_species > 0), "schnet_force_vjp: the embedding's gradient needs the species table");
    Bufs B;
    const size_t used = carve(P, true, theta, P.ws, B);
    MDG_CHECK_ARG((int64_t)used <= P.ws_floats, "schnet_force_vjp: workspace of %lld floats, %zu needed", (long long)P.ws_floats, used);
    B.gj_cap = (size_t)P.ws_floats - used;
    return run_vjp(P, B, x, w, force, dwf, theta_flat, theta, alpha, t, idx_dev, energy_colsum, stream, false, nullptr);
}
