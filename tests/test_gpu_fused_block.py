"""The fused SchNet interaction-block kernels (csrc/cfconv_fused.hip) against plain torch formulas of the same
maps (nff/nn/modules.py:531-541,564-571, nff/nn/graphconv.py:43-53 and their derivatives by torch autograd), one
kernel at a time so that a failure names the sweep, and the fused analytic path against the unfused one.
fp32 tolerances: 2e-4 relative + 2e-5 of the largest entry (MFMA f32 = k-ordered fma chains; exp2/log2 hardware
transcendentals, abs. error ~1e-7)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from test_gpu_parity import T, close, mk_system, DEV

pytestmark = pytest.mark.gpu


def _setup(G, F, seed, n_side=6, cutoff=5.0):
    from mdgrad_amd import ops, _lib
    rng = np.random.default_rng(seed)
    L = 2.9 * n_side
    g = np.stack(np.meshgrid(*[np.arange(n_side)] * 3, indexing="ij"), -1).reshape(-1, 3) * (L / n_side)
    pos = np.mod(g + rng.normal(0, 0.35, g.shape), L).astype(np.float32)
    x = T(pos, DEV)
    ell = ops.build_ell(x, _lib.make_cell(np.array([L, L, L], dtype=np.float32)), cutoff)
    topo = ops.GraphTopo(ell)
    torch.manual_seed(seed)
    mu = torch.linspace(0, cutoff, G, device=DEV)
    coef = torch.full((G,), -0.5 / float(mu[1] - mu[0]) ** 2, device=DEV)
    W1 = torch.randn(G, G, device=DEV) / G ** 0.5
    b1 = torch.randn(G, device=DEV) * 0.1
    W2 = torch.randn(F, G, device=DEV) / G ** 0.5
    b2 = torch.randn(F, device=DEV) * 0.1
    return x, topo, (mu, coef, W1, b1, W2, b2)


def _filter(d, mu, coef, W1, b1, W2, b2):
    g = torch.exp(coef * (d[:, None] - mu).pow(2))
    s = torch.nn.functional.softplus(torch.nn.functional.linear(g, W1, b1)) - np.log(2.0)
    return torch.nn.functional.linear(s, W2, b2)


def _agg(h, W, topo):
    i, j = topo.nbr[:, 0], topo.nbr[:, 1]
    return torch.zeros_like(h).index_add(0, j, h[i] * W).index_add(0, i, h[j] * W)


def _close(a, b, what):
    close(a, b, 2e-4, 2e-5 * float(b.abs().max()) + 1e-7, what)


@pytest.mark.parametrize("G,F,n_side", [(30, 128, 6), (16, 48, 6), (32, 64, 6), (41, 128, 6), (64, 32, 6), (12, 8, 6),
                                        (30, 128, 16), (30, 256, 6), (41, 512, 6), (25, 384, 6)])
def test_fused_forward_kernel_primal_and_tangent(G, F, n_side):
    from mdgrad_amd import ops
    x, topo, net = _setup(G, F, seed=G + F, n_side=n_side)
    N = topo.n_atoms
    w = torch.randn(N, 3, device=DEV)
    d, uhat, dd, ddel = ops.edge_geom(x, topo, w)
    i, j = topo.nbr[:, 0], topo.nbr[:, 1]
    delta = x[i] - x[j] - topo.offsets
    close(d, delta.norm(dim=1), 1e-6, 1e-6, "d")
    close(uhat, delta / delta.norm(dim=1)[:, None], 1e-6, 1e-6, "uhat")
    close(ddel, w[i] - w[j], 0, 1e-6, "ddel")
    close(dd, ((w[i] - w[j]) * delta).sum(1) / delta.norm(dim=1), 1e-5, 1e-5, "dd")
    fn = ops.FilterNet(*net)
    h, hd = torch.randn(N, F, device=DEV), torch.randn(N, F, device=DEV)
    Wf, Wfd = torch.autograd.functional.jvp(lambda dv: _filter(dv, *net), d, dd)
    m, md, hsum, hdsum = ops.cfconv_fwd(fn, d, None, h, None, topo, want_sums=True)
    assert md is None and hdsum is None
    _close(m, _agg(h, Wf, topo), "m (primal)")
    _close(hsum, _agg(h, torch.ones_like(Wf), topo), "hsum")
    m2, md2, hs2, hds2 = ops.cfconv_fwd(fn, d, dd, h, hd, topo, want_sums=True)
    assert torch.equal(m2, m), "the tangent variant computes the same primal bits"
    _close(md2, _agg(h, Wfd, topo) + _agg(hd, Wf, topo), "md (tangent)")
    _close(hds2, _agg(hd, torch.ones_like(Wf), topo), "hdsum")
    m3, md3, _, _ = ops.cfconv_fwd(fn, d, dd, h, None, topo)
    _close(md3, _agg(h, Wfd, topo), "md (tangent, no node tangent)")
    assert torch.equal(ops.cfconv_fwd(fn, d, dd, h, hd, topo)[1], md2), "bitwise reproducible"


def test_split_forward_sweep_equals_the_unsplit_one_to_rounding(tmp_path):
    """Up to 512 atoms the f32 forward sweep runs one atom per workgroup with its tiles dealt to the four waves (SPLIT,
    csrc/cfconv_fused.hip); MDG_FWD_SPLIT_ATOMS=0 keeps the four-atoms-per-workgroup sweep.  Same inputs in a second process
    with the variable set: every output agrees to f32 rounding (the partial rows are added in another order), and each of the
    two is reproducible bit for bit."""
    import os, subprocess, sys
    script = tmp_path / "fwd.py"
    script.write_text(
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from test_gpu_fused_block import _setup\n"
        "from test_gpu_parity import DEV\n"
        "from mdgrad_amd import ops\n"
        "x, topo, net = _setup(30, 128, seed=5)\n"
        "N = topo.n_atoms\n"
        "torch.manual_seed(1)\n"
        "w = torch.randn(N, 3, device=DEV); h = torch.randn(N, 128, device=DEV); hd = torch.randn(N, 128, device=DEV)\n"
        "d, uhat, dd, ddel = ops.edge_geom(x, topo, w)\n"
        "fn = ops.FilterNet(*net)\n"
        "a = ops.cfconv_fwd(fn, d, dd, h, hd, topo, want_sums=True)\n"
        "b = ops.cfconv_fwd(fn, d, dd, h, hd, topo, want_sums=True)\n"
        "assert all(torch.equal(p, q) for p, q in zip(a, b))\n"
        "c = ops.cfconv_fwd(fn, d, None, h, None, topo)\n"
        "assert torch.equal(c[0], a[0])\n"
        "np.savez(sys.argv[1], *[t.cpu().numpy() for t in a])\n"
        % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    outs = {}
    for tag, val in (("split", None), ("unsplit", "0")):
        env = dict(os.environ)
        env.pop("MDG_FWD_SPLIT_ATOMS", None)
        if val is not None:
            env["MDG_FWD_SPLIT_ATOMS"] = val
        out = tmp_path / (tag + ".npz")
        r = subprocess.run([sys.executable, str(script), str(out)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = np.load(out)
    worst = 0.0
    for k in outs["split"].files:
        a, b = outs["split"][k], outs["unsplit"][k]
        assert np.abs(a - b).max() <= 4e-6 * np.abs(b).max() + 1e-7, k
        worst = max(worst, float(np.abs(a - b).max()))
    assert worst > 0.0, "216 atoms: the two sweeps are expected to differ in the order of their sums (is SPLIT being taken?)"


@pytest.mark.parametrize("G,F,n_side", [(30, 128, 6), (16, 48, 6), (32, 64, 6), (41, 128, 6), (64, 32, 6), (12, 8, 6),
                                        (30, 128, 16), (30, 256, 6), (41, 512, 6), (25, 384, 6)])
def test_fused_backward_kernel_plain_dual_and_theta(G, F, n_side):
    from mdgrad_amd import ops
    x, topo, net = _setup(G, F, seed=3 * G + F, n_side=n_side)
    N, E = topo.n_atoms, topo.n_edges
    w = torch.randn(N, 3, device=DEV)
    d, uhat, dd, ddel = ops.edge_geom(x, topo, w)
    fn = ops.FilterNet(*net)
    h, hd, mb, mdb = [torch.randn(N, F, device=DEV) for _ in range(4)]
    i, j = topo.nbr[:, 0], topo.nbr[:, 1]
    Wdb = mdb[i] * h[j] + mdb[j] * h[i]
    mu, coef, W1, b1, W2, b2 = net
    for with_hd in (True, False):
        Wb = mb[i] * h[j] + mb[j] * h[i]
        if with_hd:
            Wb = Wb + mdb[i] * hd[j] + mdb[j] * hd[i]
        leaves = [t.clone().requires_grad_(True) for t in (d, dd, W1, b1, W2)]

        def S(dv, ddv, w1, bb1, w2):
            Wf, Wfd = torch.autograd.functional.jvp(lambda z: _filter(z, mu, coef, w1, bb1, w2, b2), dv, ddv,
                                                    create_graph=True)
            return (Wdb * Wfd).sum() + (Wb * Wf).sum()

        ref = torch.autograd.grad(S(*leaves), leaves)
        d_b, dd_b = torch.zeros(E, device=DEV), torch.zeros(E, device=DEV)
        th = ops.cfconv_bwd(fn, d, dd, topo, h, hd if with_hd else None, mb, mdb, d_b, dd_b, want_theta=True)
        _close(dd_b, ref[1], "dd_b (dual, hd=%s)" % with_hd)
        _close(d_b, ref[0], "d_b (dual, hd=%s)" % with_hd)
        _close(th[0], ref[2], "gW1")
        _close(th[1], ref[3], "gb1")
        _close(th[2], ref[4], "gW2")
        d_b2, dd_b2 = torch.ones(E, device=DEV), torch.ones(E, device=DEV)        # accumulate onto existing values
        assert ops.cfconv_bwd(fn, d, dd, topo, h, hd if with_hd else None, mb, mdb, d_b2, dd_b2) is None
        _close(d_b2 - 1, ref[0], "d_b (dual, no theta)")
        _close(dd_b2 - 1, ref[1], "dd_b (dual, no theta)")
    # plain reverse sweep: dU/dd with mdb in the role of dU/dm
    dleaf = d.clone().requires_grad_(True)
    (ref_d,) = torch.autograd.grad((Wdb * _filter(dleaf, *net)).sum(), dleaf)
    out = torch.zeros(E, device=DEV)
    ops.cfconv_bwd(fn, d, None, topo, h, None, None, mdb, None, out)
    _close(out, ref_d, "dU/dd (plain reverse)")
    # geometry backward
    d_b, dd_b = torch.randn(E, device=DEV), torch.randn(E, device=DEV)
    F_, dwf = ops.edge_geom_bwd(d_b, dd_b, d, dd, uhat, ddel, topo)
    sc = lambda g: torch.zeros(N, 3, device=DEV).index_add(0, i, g).index_add(0, j, -g)
    _close(F_, -sc(dd_b[:, None] * uhat), "force scatter")
    _close(dwf, -sc(d_b[:, None] * uhat + (dd_b / d)[:, None] * (ddel - dd[:, None] * uhat)), "d(w.F)/dx scatter")
    F2, none = ops.edge_geom_bwd(None, dd_b, None, None, uhat, None, topo)
    assert none is None and torch.equal(F2, F_)


@pytest.mark.parametrize("G,F,n_side", [(30, 128, 6), (16, 48, 6), (41, 128, 6), (30, 256, 6), (30, 128, 16), (25, 384, 6)])
@pytest.mark.parametrize("mode", ["f32", "bf16", "rows16"])
def test_bias_gradient_from_the_spare_filter_column(G, F, n_side, mode):
    """mdg_cfconv_bwd_theta's gb2 = d/d b2 of  sum_e Wdb_e . Wd_e + Wb_e . W_e  =  sum_e Wb_e  (W = s W2 + b2, Wd has no bias):
    the padded column of the first filter layer with its activation pinned to 1.  Against the sum formed in torch, with the
    other gradients unchanged by the column (f32: to rounding; bf16 operands: Wb is rounded to bf16 before it is summed)."""
    from mdgrad_amd import ops
    x, topo, net = _setup(G, F, seed=5 * G + F, n_side=n_side)
    N, E = topo.n_atoms, topo.n_edges
    w = torch.randn(N, 3, device=DEV)
    d, uhat, dd, ddel = ops.edge_geom(x, topo, w)
    fn = ops.FilterNet(*net, bf16=mode != "f32", rows16=mode == "rows16")
    if mode == "rows16" and not fn.rows16:
        pytest.skip("bf16 node rows need more than 64 filters")
    assert fn.b2col
    h, hd, mb, mdb = [torch.randn(N, F, device=DEV) for _ in range(4)]
    if mode == "rows16":
        h16, hd16, mb16, mdb16 = [ops.rows_to_bf16(v) for v in (h, hd, mb, mdb)]
        h, hd, mb, mdb = [v.float() for v in (h16, hd16, mb16, mdb16)]
    i, j = topo.nbr[:, 0], topo.nbr[:, 1]
    for with_hd in (True, False):
        Wb = mb[i] * h[j] + mb[j] * h[i]
        if with_hd:
            Wb = Wb + mdb[i] * hd[j] + mdb[j] * hd[i]
        ref = Wb.double().sum(0).float()
        args = (h16, hd16 if with_hd else None, mb16, mdb16) if mode == "rows16" else (h, hd if with_hd else None, mb, mdb)
        d_b, dd_b = torch.zeros(E, device=DEV), torch.zeros(E, device=DEV)
        th = ops.cfconv_bwd(fn, d, dd, topo, *args, d_b, dd_b, want_theta=True, want_b2=True)
        d_b2, dd_b2 = torch.zeros(E, device=DEV), torch.zeros(E, device=DEV)
        th0 = ops.cfconv_bwd(fn, d, dd, topo, *args, d_b2, dd_b2, want_theta=True)
        assert len(th) == 4 and len(th0) == 3
        scale = float(Wb.abs().sum(0).max())                        # (a sum of E signed terms: errors relative to its terms)
        # (f32: the partial sums of ~10^3..10^5 terms in f32, observed 2.4e-6; bf16 operands: 2^-9 per rounded term)
        close(th[3], ref, 0, (1e-5 if mode == "f32" else 3e-3) * scale, "gb2 (%s, hd=%s)" % (mode, with_hd))
        for a, b in zip(th[:3], th0):
            assert torch.equal(a, b), "the other parameter gradients do not see the column"
        assert torch.equal(d_b, d_b2) and torch.equal(dd_b, dd_b2)
    full = _setup(32, F, seed=2, n_side=n_side)[2]
    assert not ops.FilterNet(*full).b2col, "no spare column at n_gaussians = 32"


def test_fused_kernels_on_a_padded_fixed_capacity_topology():
    """StaticTopo (HIP-graph capture): padding rows (-1, -1) are inert in the edge-centric kernel."""
    from mdgrad_amd import ops
    G, F = 30, 128
    x, topo, net = _setup(G, F, seed=5)
    need = torch.zeros(2, dtype=torch.int32, device=DEV)
    st = ops.StaticTopo(topo.ell, topo.n_edges + 777, need)
    assert st.n_edges == topo.n_edges + 777
    N = topo.n_atoms
    w = torch.randn(N, 3, device=DEV)
    fn = ops.FilterNet(*net)
    h, hd, mb, mdb = [torch.randn(N, F, device=DEV) for _ in range(4)]
    outs = []
    for tp in (topo, st):
        d, uhat, dd, ddel = ops.edge_geom(x, tp, w)
        m = ops.cfconv_fwd(fn, d, dd, h, hd, tp)
        d_b, dd_b = torch.zeros(tp.n_edges, device=DEV), torch.zeros(tp.n_edges, device=DEV)
        th = ops.cfconv_bwd(fn, d, dd, tp, h, hd, mb, mdb, d_b, dd_b, want_theta=True)
        F_, dwf = ops.edge_geom_bwd(d_b, dd_b, d, dd, uhat, ddel, tp)
        outs.append([m[0], m[1], d_b[:topo.n_edges], dd_b[:topo.n_edges], F_, dwf] + list(th))
        if tp is st:
            assert float(d_b[topo.n_edges:].abs().max()) == 0.0 and float(dd_b[topo.n_edges:].abs().max()) == 0.0
    for k, (a, b) in enumerate(zip(*outs)):
        if k < 6:
            assert torch.equal(a, b), "padded topology output %d" % k
        else:                                    # partial sums are grouped by tile, so the split over workgroups differs
            _close(b, a, "padded topology theta %d" % k)


@pytest.mark.parametrize("A,F,G", [(512, 512, 41), (128, 384, 12)])
def test_widest_search_space_settings_run_on_the_fused_kernels_vs_autograd(A, F, G):
    """demo/fit_rdf_gnn.py:16-19: n_atom_basis / n_filters up to 512.  The fused block takes them (filters in chunks of
    128, Dense layers in k-slabs of 256): force, d(w.F)/dx and the 1.9 M-entry d(w.F)/dtheta against the autograd path
    (double backward through SchNet.forward) on a 64-bead box."""
    from mdgrad_amd import ops
    from mdgrad_amd.interface import GNNPotentials
    from mdgrad_amd.nn import get_model, analytic
    assert ops.FilterNet.supported(G, F)
    g = load_golden("schnet_cg64_wide")
    system = mk_system(g["pos"], g["cell"], mass=g["masses"], numbers=g["numbers"])
    torch.manual_seed(A + F)
    net = get_model({"n_atom_basis": A, "n_filters": F, "n_gaussians": G, "n_convolutions": 2, "cutoff": 6.0})
    gnn = GNNPotentials(system, net, cutoff=6.0)
    q = T(g["pos"], DEV)
    gnn._reset_topology(q)
    assert analytic.fused_ok(net)
    w = T(np.random.default_rng(2).normal(0, 1, g["pos"].shape).astype(np.float32), DEV)
    U, F_, dq, gth = analytic.force_vjp(net, gnn._z(), q, w, gnn.inputs["_topo"])
    qa = q.clone().requires_grad_(True)
    Ua = gnn(qa).sum()
    (gq,) = torch.autograd.grad(Ua, qa, create_graph=True)
    plist = list(net.parameters())
    ga = torch.autograd.grad((w * -gq).sum(), [qa] + plist, allow_unused=True)
    close(U.reshape(1), Ua.detach().reshape(1), 1e-4, 1e-4, "U")
    close(F_, -gq.detach(), 1e-3, 1e-4 * float(gq.abs().max()), "F")
    close(dq, ga[0], 2e-3, 2e-4 * float(ga[0].abs().max()), "d(w.F)/dx")
    flat = torch.cat([t.reshape(-1) for t in gth])
    fa = torch.cat([(x if x is not None else torch.zeros_like(p)).reshape(-1) for x, p in zip(ga[1:], plist)])
    close(flat, fa, 2e-3, 2e-4 * float(fa.abs().max()), "d(w.F)/dtheta")
    # bf16 filter operands at the same widths: within the stated bf16 tolerance of the f32 path
    net.filter_bf16 = True
    _, F16, dq16, g16 = analytic.force_vjp(net, gnn._z(), q, w, gnn.inputs["_topo"])
    close(F16, F_, 0, 2e-2 * float(F_.abs().max()), "F (bf16 filter)")
    f16 = torch.cat([t.reshape(-1) for t in g16])
    assert float((f16.double() * flat.double()).sum() / (f16.double().norm() * flat.double().norm())) > 0.999


@pytest.mark.parametrize("name", ["schnet_cg64", "schnet_water192", "schnet_cg64_wide", "schnet_cg64_a256"])
def test_fused_analytic_path_equals_unfused(name):
    from mdgrad_amd.interface import GNNPotentials
    from mdgrad_amd.nn import get_model, analytic
    from test_gpu_schnet import params_of, sd_of
    g = load_golden(name)
    system = mk_system(g["pos"], g["cell"], mass=g["masses"], numbers=g["numbers"])
    net = get_model(params_of(g))
    net.load_state_dict(sd_of(g))
    gnn = GNNPotentials(system, net, cutoff=float(g["cutoff"]))
    q = T(g["pos"], DEV)
    gnn._reset_topology(q)
    assert analytic.fused_ok(net)
    rng = np.random.default_rng(1)
    w = T(rng.normal(0, 1, g["pos"].shape).astype(np.float32), DEV)
    res = []
    for fused in (True, False):
        net.fused_block = fused
        U, F = analytic.force(net, gnn._z(), q, gnn.inputs["_topo"])
        U2, F2, dq, gth = analytic.force_vjp(net, gnn._z(), q, w, gnn.inputs["_topo"])
        _, F3, dq3, none = analytic.force_vjp(net, gnn._z(), q, w, gnn.inputs["_topo"], want_theta=False)
        assert none is None
        res.append([U.reshape(1), F, U2.reshape(1), F2, dq, F3, dq3, torch.cat([t.reshape(-1) for t in gth])])
    for k, (a, b) in enumerate(zip(*res)):
        close(a, b, 1e-4, 2e-5 * float(b.abs().max()) + 1e-7, "fused vs unfused #%d" % k)
    close(res[0][1], g["F"], 1e-4, 1e-5 * np.abs(g["F"]).max(), "fused F vs golden")


@pytest.mark.parametrize("N,K,M", [(1000, 64, 128), (4096, 128, 64), (37, 30, 30), (513, 200, 130), (64, 32, 1), (300, 64, 64),
                                   (700, 512, 256), (300, 512, 512), (129, 300, 512), (200, 770, 64)])
def test_dense_node_kernel_epilogues(N, K, M):
    """csrc/dense.hip against torch: linear / transposed weight, bias, shifted softplus with its tangent row,
    elementwise product and residuals, single and dual inputs."""
    from mdgrad_amd import ops
    torch.manual_seed(N + K + M)
    x0, x1 = torch.randn(N, K, device=DEV), torch.randn(N, K, device=DEV)
    W = torch.randn(M, K, device=DEV) / K ** 0.5
    bias = torch.randn(M, device=DEV)
    mul, res, res1 = [torch.randn(N, M, device=DEV) for _ in range(3)]
    ln2 = float(np.log(2.0))
    out0, sig, out1 = ops.dense(W, x0, bias=bias, act=True, x1=x1, res1=res1, want_sig=True)
    z = x0 @ W.t() + bias
    _close(out0, torch.nn.functional.softplus(z) - ln2, "ssp(x W^T + b)")
    _close(sig, torch.sigmoid(z), "sigmoid")
    _close(out1, torch.sigmoid(z) * (x1 @ W.t()) + res1, "tangent row")
    out0, sig, out1 = ops.dense(W, x0, bias=bias, res=res, x1=x1)
    assert sig is None
    _close(out0, z + res, "x W^T + b + res")
    _close(out1, x1 @ W.t(), "second row, no residual")
    Wt = W.t().contiguous()                                    # [K, M]: trans = True contracts over the first index
    out0, _, out1 = ops.dense(Wt, x0, trans=True, mul=mul, x1=x1, res1=res1)
    _close(out0, (x0 @ Wt) * mul, "(x W) * mul")
    _close(out1, x1 @ Wt + res1, "x1 W + res1")
    single = ops.dense(W, x0, bias=bias)[0]
    _close(single, z, "single input")
    assert torch.equal(single, ops.dense(W, x0, bias=bias)[0])


def test_large_system_eager_pass_on_fixed_capacity_lists_equals_exact_lists():
    """Systems above graphs.MAX_EDGES run eagerly but on fixed-capacity (sync-free) neighbour lists: same trajectory
    and gradients as the plain eager path on exact-size lists; an undersized capacity is detected and the pass redone."""
    from mdgrad_amd import graphs
    from test_gpu_schnet import _gnn_integrator, _traj_and_grads
    g = load_golden("gnn_traj")
    t = torch.Tensor([float(g["dt"]) * i for i in range(6)]).to(DEV)
    res = []
    saved = graphs.MAX_EDGES
    try:
        for mode in ("exact", "static", "static-overflow"):
            system = mk_system(g["pos"], g["cell"], g["vel"], g["masses"], g["numbers"])
            integ = _gnn_integrator(g, system)
            if mode == "exact":
                integ.use_graphs = False
            else:
                graphs.MAX_EDGES = 0                              # every system counts as "too large for replay"
                if mode == "static-overflow":
                    integ.model.set_static_topology(True)
                    for m in integ.model.models.values():         # capacities far too small: must be detected
                        m._static["max_nbr"] = 8
                        if "capacity" in m._static:
                            m._static["capacity"] = 64
                    integ.model.set_static_topology(False)
            res.append(_traj_and_grads(integ, system, t))
            assert not any(m._static_on for m in integ.model.models.values()), "exact-size lists restored"
    finally:
        graphs.MAX_EDGES = saved
    for k in (1, 2):
        for a, b, nm in zip(res[k], res[0], ("v_t", "q_t", "pv_t", "dL/dtheta")):
            close(a, b, 1e-4, 2e-5 * float(b.abs().max()) + 1e-7, "%s, mode %d" % (nm, k))


def test_atb2_kernel():
    """mdg_atb2: a^T b + a2^T b2 in one split-K launch pair -- against torch."""
    from mdgrad_amd import ops
    torch.manual_seed(5)
    for rows, m, n in ((4096, 64, 128), (32768, 128, 64), (300, 30, 7)):
        A, A2 = torch.randn(rows, m, device=DEV), torch.randn(rows, m, device=DEV)
        B, B2 = torch.randn(rows, n, device=DEV), torch.randn(rows, n, device=DEV)
        ref = (A.double().t() @ B.double() + A2.double().t() @ B2.double()).float()
        _close(ops._atb2(A, B, A2, B2), ref, "atb2")
        assert torch.equal(ops._atb2(A, B, A2, B2), ops._atb2(A, B, A2, B2)), "bitwise reproducible"


def test_grad_jobs_batched_parameter_gradient_reductions():
    """mdg_grad_jobs (csrc/gradjobs.hip): tall-skinny products (one / two operand pairs, odd widths, a single-column
    operand with re-mapped destination rows), column sums (plain, of a product, of two products), pre-reduced pieces --
    35 jobs in two chunks -- accumulated into a flat buffer with the interval weight read from a device time grid;
    against fp64 torch.  Two runs give the same bits."""
    from mdgrad_amd import ops
    torch.manual_seed(3)
    N = 5000
    rnd = lambda *s: torch.randn(*s, device=DEV)
    shapes = [(64, 128), (32, 64), (128, 64), (64, 64), (30, 66), (100, 36), (3, 64), (256, 256)]
    params = [torch.nn.Parameter(torch.zeros(m, n, device=DEV)) for m, n in shapes]
    params += [torch.nn.Parameter(torch.zeros(c, device=DEV)) for c in (128, 64, 30, 1, 257)]
    emb = torch.nn.Parameter(torch.zeros(100, 64, device=DEV))
    pre = [torch.nn.Parameter(torch.zeros(30, 30, device=DEV)), torch.nn.Parameter(torch.zeros(30, device=DEV))]
    more = [torch.nn.Parameter(torch.zeros(30, device=DEV)) for _ in range(20)]      # (a destination appears once per call)
    allp = params + [emb] + pre + more
    t = torch.tensor([0.0, 0.3, 0.7, 1.5], device=DEV)
    idx = torch.tensor([2], dtype=torch.int64, device=DEV)
    outs = []
    for rep in range(2):
        acc = ops.ThetaAccum(allp, t=t, idx=idx)
        acc.flat.copy_(torch.arange(acc.n, device=DEV, dtype=torch.float32) * 1e-3)
        base = acc.flat.double().clone()
        torch.manual_seed(4)
        jobs, want = ops.GradJobs(), {}
        for k, (p, (m, n)) in enumerate(zip(params, shapes)):
            A, B = rnd(N, m), rnd(N, n)
            if k % 2 == 0:
                A2, B2 = rnd(N, m), rnd(N, n)
                jobs.atb(acc.off[id(p)], A, B, A2, B2)
                want[id(p)] = A.double().t() @ B.double() + A2.double().t() @ B2.double()
            else:
                jobs.atb(acc.off[id(p)], A, B)
                want[id(p)] = A.double().t() @ B.double()
        for k, p in enumerate(params[len(shapes):]):
            c = p.numel()
            A, B, A2, B2 = rnd(N, c), rnd(N, c), rnd(N, c), rnd(N, c)
            if k % 3 == 0:
                jobs.colsum(acc.off[id(p)], A)
                want[id(p)] = A.double().sum(0)
            elif k % 3 == 1:
                jobs.colsum(acc.off[id(p)], A, B)
                want[id(p)] = (A.double() * B.double()).sum(0)
            else:
                jobs.colsum(acc.off[id(p)], A, B, A2, B2)
                want[id(p)] = (A.double() * B.double() + A2.double() * B2.double()).sum(0)
        uniq = torch.tensor([1, 8, 17], dtype=torch.int64, device=DEV)
        onehot = torch.nn.functional.one_hot(torch.randint(0, 3, (N,), device=DEV), 3).float()
        rb = rnd(N, 64)
        jobs.atb(acc.off[id(emb)], onehot, rb, row_map=uniq)
        w_emb = torch.zeros(100, 64, device=DEV, dtype=torch.float64)
        w_emb[uniq] = onehot.double().t() @ rb.double()
        want[id(emb)] = w_emb
        for p in pre:
            g = rnd(*p.shape)
            jobs.axpy(acc.off[id(p)], g)
            want[id(p)] = g.double()
        for p in more:                                            # second chunk (> 32 jobs per evaluation)
            g = rnd(30)
            jobs.axpy(acc.off[id(p)], g)
            want[id(p)] = g.double()
        assert len(jobs.jobs) > 32
        jobs.run(acc, alpha=-1.0, accumulate=True)
        ref = base.clone()
        dt = float(t[2] - t[1])
        for p in allp:
            o = acc.off[id(p)]
            ref[o:o + p.numel()] += -dt * want[id(p)].reshape(-1)
        scale = float(ref.abs().max())
        err = float((acc.flat.double() - ref).abs().max())
        assert err < 2e-5 * scale, "grad jobs vs fp64: %.3e (scale %.3e)" % (err, scale)
        # without accumulation the destination is overwritten
        acc2 = ops.ThetaAccum(pre[1:])
        acc2.flat.fill_(7.0)
        j2 = ops.GradJobs()
        g = rnd(30)
        j2.axpy(0, g)
        j2.run(acc2, alpha=2.0, accumulate=False)
        assert torch.allclose(acc2.flat, 2.0 * g)
        outs.append(acc.flat.clone())
    assert torch.equal(outs[0], outs[1]), "fixed-order reductions: two runs must give the same bits"


def test_readout_head_kernel():
    from mdgrad_amd import ops
    torch.manual_seed(0)
    sy, syd, L2 = torch.rand(777, 32, device=DEV), torch.randn(777, 32, device=DEV), torch.randn(1, 32, device=DEV)
    ydb, yb = ops.readout_head(sy, syd, L2)
    assert torch.allclose(ydb, sy * L2) and torch.allclose(yb, (1 - sy) * syd * L2, atol=1e-6)
    ydb1, none = ops.readout_head(sy, None, L2)
    assert none is None and torch.equal(ydb1, ydb)


@pytest.mark.parametrize("bf16", [False, True])
def test_trainable_gaussian_basis_on_the_fused_kernels_vs_autograd(bf16):
    """GaussianSmearing(trainable=True) (nff/nn/layers.py:34-83; `trainable_gauss` of demo/fit_rdf_gnn.py): `offsets` and `width`
    are parameters.  The fused reverse sweep returns their gradients too (mdg_cfconv_bwd_smear): force, d(w.F)/dx and every
    d(w.F)/dtheta -- the 2 x 30 basis parameters of each layer included -- against the autograd path (double backward through
    SchNet.forward); a trajectory through the analytic adjoint then matches the generic autograd adjoint."""
    from mdgrad_amd.interface import GNNPotentials
    from mdgrad_amd.nn import get_model, analytic
    g = load_golden("schnet_cg64_wide")
    system = mk_system(g["pos"], g["cell"], mass=g["masses"], numbers=g["numbers"])
    torch.manual_seed(21)
    net = get_model({"n_atom_basis": 64, "n_filters": 128, "n_gaussians": 30, "n_convolutions": 2, "cutoff": 6.0,
                     "trainable_gauss": True})
    with torch.no_grad():                       # (move the basis off its initial grid so that nothing is symmetric by accident)
        for conv in net.convolutions:
            sm = conv.moduledict["message_edge_filter"][0]
            sm.offsets.add_(torch.randn_like(sm.offsets) * 0.03)
            sm.width.mul_(1.0 + 0.1 * torch.rand_like(sm.width))
    names = [n for n, _ in net.named_parameters()]
    assert any(n.endswith("message_edge_filter.0.width") for n in names) and any(n.endswith("message_edge_filter.0.offsets") for n in names)
    gnn = GNNPotentials(system, net, cutoff=6.0)
    assert analytic.supported(net) and analytic.fused_ok(net) and gnn.supports_force_vjp()
    q = T(g["pos"], DEV)
    gnn._reset_topology(q)
    w = T(np.random.default_rng(4).normal(0, 1, g["pos"].shape).astype(np.float32), DEV)
    qa = q.clone().requires_grad_(True)
    Ua = gnn(qa).sum()
    (gq,) = torch.autograd.grad(Ua, qa, create_graph=True)
    plist = list(net.parameters())
    ga = torch.autograd.grad((w * -gq).sum(), [qa] + plist, allow_unused=True)
    net.filter_bf16 = bf16
    U, F_, dq, gth = analytic.force_vjp(net, gnn._z(), q, w, gnn.inputs["_topo"])
    tol = 2e-4 if not bf16 else 2e-2
    close(F_, -gq.detach(), 0, tol * float(gq.abs().max()), "F")
    close(dq, ga[0], 0, tol * float(ga[0].abs().max()), "d(w.F)/dx")
    for (n, p), got, ref in zip(net.named_parameters(), gth, ga[1:]):
        ref = ref if ref is not None else torch.zeros_like(p)
        if n.endswith(".0.width") or n.endswith(".0.offsets") or not bf16:
            close(got, ref, 0, (5e-4 if not bf16 else 3e-2) * float(ref.abs().max()) + 1e-7, "d(w.F)/d " + n)
    fa = torch.cat([(x if x is not None else torch.zeros_like(p)).reshape(-1) for x, p in zip(ga[1:], plist)])
    flat = torch.cat([t.reshape(-1) for t in gth])
    cos = float((flat.double() * fa.double()).sum() / (flat.double().norm() * fa.double().norm()))
    assert cos > (0.999999 if not bf16 else 0.9999), cos


@pytest.mark.parametrize("N,K,M1,M2,M3", [(1000, 128, 64, 64, 32), (4096, 64, 128, 128, 64), (37, 50, 25, 25, 12),
                                          (530, 512, 512, 512, 256), (129, 384, 130, 130, 65), (16, 8, 8, 8, 4),
                                          # more than 8 192 rows: the many-row variants of the kernel (no operand prefetch,
                                          # three / four waves per SIMD), which the stacked 8 x 4 096-bead workload runs on
                                          (9001, 128, 64, 64, 32), (8200, 384, 130, 130, 65), (32768, 64, 128, 128, 64), (16400, 128, 64, 64, 32),
                                          # the other widths the compiled chains exist for (A = F = 128, A = F = 64, A = 128 / F = 64)
                                          (777, 128, 128, 128, 64), (555, 64, 64, 64, 32), (640, 64, 128, 128, 64)])
@pytest.mark.parametrize("walker", [False, True])
def test_row_chain_kernel_every_stage_vs_torch(N, K, M1, M2, M3, walker, monkeypatch):
    """csrc/rowchain.hip: the stretch of the SchNet sweeps around the readout as ONE launch -- update MLP (activation,
    sigmoid, tangent), residual, readout with the head transform, then the three transposed layers with the reverse of the
    (ssp, tangent) pair -- every saved tensor against the same sequence in torch, dual and single rows, widths that are
    not multiples of 16 / 4 and a row count that is not a multiple of the 16-row tile."""
    from mdgrad_amd import ops, _lib
    # (F, A, A, A / 2) = (128, 64, 64, 32), (128, 128, 128, 64) and (64, 64, 64, 32) have the shape of the SchNet turn chain: the
    # library runs its compiled version of it (csrc/rowchain.hip, "specialised chains"); walker = True forces the
    # descriptor walker on the same lists, so both are pinned to the same torch sequence
    if walker:
        monkeypatch.setenv("MDG_CHAIN_WALKER", "1")
    else:
        monkeypatch.delenv("MDG_CHAIN_WALKER", raising=False)
    torch.manual_seed(N + K + M1)
    rn = lambda *s: torch.randn(*s, device=DEV)
    W1, W2, W3 = rn(M1, K) / K ** 0.5, rn(M2, M1) / M1 ** 0.5, rn(M3, M2) / M2 ** 0.5
    b1, b2, b3, l = rn(M1), rn(M2), rn(M3), rn(1, M3)
    x0, x1, r0, r1 = rn(N, K), rn(N, K), rn(N, M2), rn(N, M2)
    ln2 = float(np.log(2.0))
    ssp = lambda z: torch.nn.functional.softplus(z) - ln2
    for dual in (True, False):
        ch = ops.RowChain(N, dual, x0.device)
        a = ch.stage(W1, bias=b1, act=True, in0=x0, in1=x1 if dual else None, want_sig=True)
        b = ch.stage(W2, bias=b2, res0=r0, res1=r1 if dual else None)
        y = ch.stage(W3, bias=b3, act=True, mode=_lib.CHAIN_HEAD, aux0=l, want_sig=True, want_pre=(True, True))
        g = ch.stage(W3, trans=True)
        if dual:
            e = ch.stage(W2, trans=True, mode=_lib.CHAIN_SSP_BWD, aux0=a.sig, aux1=a.out1)
        else:
            e = ch.stage(W2, trans=True, mode=_lib.CHAIN_MUL, aux0=a.sig)
        f = ch.stage(W1, trans=True)
        ch.run()
        z1 = x0 @ W1.t() + b1
        t, su = ssp(z1), torch.sigmoid(z1)
        r = t @ W2.t() + b2 + r0
        z3 = r @ W3.t() + b3
        sy = torch.sigmoid(z3)
        ydb = sy * l
        rdb = ydb @ W3
        tdb = rdb @ W2
        _close(a.out0, t, "t"); _close(a.sig, su, "su"); _close(b.out0, r, "r"); _close(y.sig, sy, "sy")
        _close(y.pre0, ssp(z3), "ty"); _close(y.out0, ydb, "ydb"); _close(g.out0, rdb, "rdb")
        _close(e.out0, su * tdb, "udb"); _close(f.out0, (su * tdb) @ W1, "mdb")
        if dual:
            td = su * (x1 @ W1.t())
            rd = td @ W2.t() + r1
            syd = sy * (rd @ W3.t())
            yb = (1 - sy) * syd * l
            rb = yb @ W3
            tb = rb @ W2
            ub = (1 - su) * td * tdb + su * tb
            _close(a.out1, td, "td"); _close(b.out1, rd, "rd"); _close(y.pre1, syd, "syd"); _close(y.out1, yb, "yb")
            _close(g.out1, rb, "rb"); _close(e.out1, ub, "ub"); _close(f.out1, ub @ W1, "mb")
        else:
            assert a.out1 is None and y.pre1 is None and f.out1 is None
    # a stage whose copy is not asked for leaves no tensor, and the chain still feeds the next stage
    ch = ops.RowChain(N, False, x0.device)
    ch.stage(W1, bias=b1, in0=x0, store=False)
    o = ch.stage(W2, bias=b2)
    ch.run()
    _close(o.out0, (x0 @ W1.t() + b1) @ W2.t() + b2, "two plain stages, first one not stored")
    with pytest.raises(RuntimeError):
        bad = ops.RowChain(N, False, x0.device)
        bad.stage(W1, in0=x0)
        bad.stage(rn(5, M1 + 1))                                   # k = M1 + 1 does not match the previous width M1
        bad.run()


@pytest.mark.parametrize("x3", [0, 2, 4])
@pytest.mark.parametrize("N", [1000, 16400 + 7, 32768])
@pytest.mark.parametrize("A,F", [(64, 128), (64, 64), (128, 128)])
def test_row_chain_looping_and_split_bf16_variants_vs_torch(A, F, N, x3):
    """The three compiled chains of an n_atom_basis = 64 network (forward: update MLP + residual + next node filter with its
    bf16 mirrors; turn; reverse) in the variants round 6 added: LOOP (>= 1 024 row tiles: two workgroups per CU keep the
    weight fragments in registers and walk the tiles, the next tile's rows prefetched; the last tile ragged) and
    MDG_CHAIN_X3 (x3 = 2: products as three bf16 MFMAs on split operands, ~1e-5 relative) and MDG_CHAIN_X6 (x3 = 4: six products
    of three exact bf16 pieces per operand, the f32 matrix instruction's accuracy) -- every output against torch in float64.
    (A = 128: no LOOP, X3 not compiled -- the flag falls back to f32 products --, X6 on up to 398 registers per lane.)"""
    from mdgrad_amd import ops, _lib
    torch.manual_seed(N + A + F + int(x3))
    rn = lambda *s: torch.randn(*s, device=DEV)
    H = A // 2
    U1, c1, U2, c2 = rn(A, F) / F ** 0.5, rn(A), rn(A, A) / A ** 0.5, rn(A)
    Wn, bn, L1, l1, L2 = rn(F, A) / A ** 0.5, rn(F), rn(H, A) / A ** 0.5, rn(H), rn(1, H)
    m, md, r, rd = rn(N, F), rn(N, F), rn(N, A), rn(N, A)
    D = lambda t: t.double()
    ln2 = float(np.log(2.0))
    ssp = lambda z: torch.nn.functional.softplus(z) - ln2
    tol = 3e-4 if (x3 == 2 and A == 64) else 2e-5

    def near(a, b, what):
        b = b.float()
        err = float((a.float() - b).abs().max())
        assert err <= tol * float(b.abs().max()) + 1e-6, (what, err, float(b.abs().max()))

    for dual in (True, False):
        # forward chain
        ch = ops.RowChain(N, dual, DEV, x3)
        a = ch.stage(U1, bias=c1, act=True, in0=m, in1=md if dual else None, want_sig=True)
        b = ch.stage(U2, bias=c2, res0=r, res1=rd if dual else None)
        c = ch.stage(Wn, bias=bn, mirror=True)
        ch.run()
        z1 = D(m) @ D(U1).t() + D(c1)
        t, su = ssp(z1), torch.sigmoid(z1)
        rn_ = t @ D(U2).t() + D(c2) + D(r)
        hn = rn_ @ D(Wn).t() + D(bn)
        near(a.out0, t, "t"); near(a.sig, su, "su"); near(b.out0, rn_, "r'"); near(c.out0, hn, "h'")
        assert torch.equal(c.out0_h, c.out0.to(torch.bfloat16)), "mirror of h'"
        if dual:
            td = su * (D(md) @ D(U1).t())
            rdn = td @ D(U2).t() + D(rd)
            near(a.out1, td, "td"); near(b.out1, rdn, "rd'"); near(c.out1, rdn @ D(Wn).t(), "hd'")
            assert torch.equal(c.out1_h, c.out1.to(torch.bfloat16))
        # turn chain
        ch = ops.RowChain(N, dual, DEV, x3)
        a = ch.stage(U1, bias=c1, act=True, in0=m, in1=md if dual else None, want_sig=True)
        b = ch.stage(U2, bias=c2, res0=r, res1=rd if dual else None)
        y = ch.stage(L1, bias=l1, act=True, mode=_lib.CHAIN_HEAD, aux0=L2, want_sig=True, want_pre=(True, True))
        g = ch.stage(L1, trans=True)
        e = ch.stage(U2, trans=True, mode=_lib.CHAIN_SSP_BWD if dual else _lib.CHAIN_MUL, aux0=a.sig, aux1=a.out1 if dual else None)
        f = ch.stage(U1, trans=True, mirror=True)
        ch.run()
        z3 = rn_ @ D(L1).t() + D(l1)
        sy = torch.sigmoid(z3)
        ydb = sy * D(L2)
        rdb = ydb @ D(L1)
        tdb = rdb @ D(U2)
        near(y.pre0, ssp(z3), "ty"); near(y.sig, sy, "sy"); near(g.out0, rdb, "rdb"); near(e.out0, su * tdb, "udb")
        near(f.out0, (su * tdb) @ D(U1), "mdb")
        assert torch.equal(f.out0_h, f.out0.to(torch.bfloat16))
        if dual:
            syd = sy * (rdn @ D(L1).t())
            yb = (1 - sy) * syd * D(L2)
            rb = yb @ D(L1)
            ub = (1 - su) * td * tdb + su * (rb @ D(U2))
            near(y.pre1, syd, "syd"); near(g.out1, rb, "rb"); near(e.out1, ub, "ub"); near(f.out1, ub @ D(U1), "mb")
        # reverse chain (adjoint rows hb / hdb of the next block's filtered rows coming in)
        hb, hdb, gb, gdb = rn(N, F), rn(N, F), rn(N, A), rn(N, A)
        ch = ops.RowChain(N, dual, DEV, x3)
        g2 = ch.stage(Wn, trans=True, in0=hb, in1=hdb if dual else None, res0=gb, res1=gdb if dual else None)
        e2 = ch.stage(U2, trans=True, mode=_lib.CHAIN_SSP_BWD if dual else _lib.CHAIN_MUL, aux0=a.sig, aux1=a.out1 if dual else None)
        f2 = ch.stage(U1, trans=True)
        ch.run()
        # (the saved activation rows a.sig / a.out1 are the kernel's own: the reference uses them as they are)
        su_k, td_k = D(a.sig), (D(a.out1) if dual else None)
        q0 = D(hb) @ D(Wn) + D(gb)
        t0 = q0 @ D(U2)
        near(g2.out0, q0, "rev r"); near(e2.out0, su_k * t0, "rev u"); near(f2.out0, (su_k * t0) @ D(U1), "rev m")
        if dual:
            q1 = D(hdb) @ D(Wn) + D(gdb)
            u1 = (1 - su_k) * td_k * t0 + su_k * (q1 @ D(U2))
            near(g2.out1, q1, "rev rd"); near(e2.out1, u1, "rev ud"); near(f2.out1, u1 @ D(U1), "rev md")


@pytest.mark.parametrize("A,F,G,convs,R", [(64, 128, 30, 2, 1), (48, 64, 16, 3, 1), (100, 96, 25, 1, 1), (64, 128, 30, 2, 160)])
def test_row_chain_path_equals_layer_by_layer_path(A, F, G, convs, R):
    """analytic.force / force_vjp with the node-level layers chained (one launch per stretch between aggregations) against
    the layer-by-layer launches of csrc/dense.hip: energy, force, d(w.F)/dx, d(w.F)/dtheta -- 1, 2 and 3 interaction
    blocks (the inner forward / reverse chains only exist from 2 blocks on), widths that are not multiples of 16."""
    from mdgrad_amd.interface import GNNPotentials
    from mdgrad_amd.nn import get_model, analytic
    g = load_golden("schnet_cg64")
    system = mk_system(g["pos"], g["cell"], mass=g["masses"], numbers=g["numbers"])
    pos = g["pos"]
    if R > 1:            # R stacked replicas (10 240 rows at R = 160: the many-row variants of the chain kernel)
        system = system.replicate(R)
        rng = np.random.default_rng(17)
        pos = np.concatenate([np.mod(g["pos"] + rng.normal(0, 0.03, g["pos"].shape), g["cell"]) for _ in range(R)]).astype(np.float32)
        system.set_positions(pos)
    torch.manual_seed(A + convs)
    net = get_model({"n_atom_basis": A, "n_filters": F, "n_gaussians": G, "n_convolutions": convs, "cutoff": 6.0})
    gnn = GNNPotentials(system, net, cutoff=6.0)
    q = T(pos, DEV)
    gnn._reset_topology(q)
    assert analytic.chain_ok(net)
    w = T(np.random.default_rng(3).normal(0, 1, pos.shape).astype(np.float32), DEV)
    res = []
    for chain in (True, False):
        net.row_chain = chain
        assert analytic.chain_ok(net) == chain
        U, F_ = analytic.force(net, gnn._z(), q, gnn.inputs["_topo"])
        U2, F2, dq, gth = analytic.force_vjp(net, gnn._z(), q, w, gnn.inputs["_topo"])
        Un, F3, dq3, none = analytic.force_vjp(net, gnn._z(), q, w, gnn.inputs["_topo"], want_theta=False, want_energy=False)
        assert none is None and Un is None
        res.append([U.reshape(1), F_, U2.reshape(1), F2, dq, F3, dq3, torch.cat([t.reshape(-1) for t in gth])])
    for k, (a, b) in enumerate(zip(*res)):
        close(a, b, 1e-4, 2e-5 * float(b.abs().max()) + 1e-7, "chained vs layer-by-layer #%d" % k)
    # the transposed weight copies the reverse stages read (analytic._wt) follow an in-place update of the weights
    with torch.no_grad():
        for conv in net.convolutions:
            conv.moduledict["update_function"][2].weight.mul_(1.25)
            conv.moduledict["message_node_filter"].weight.add_(0.01)
        net.atomwisereadout.readout["energy"][0].weight.mul_(0.8)
    res = []
    for chain in (True, False):
        net.row_chain = chain
        U2, F2, dq, gth = analytic.force_vjp(net, gnn._z(), q, w, gnn.inputs["_topo"])
        res.append([F2, dq, torch.cat([t.reshape(-1) for t in gth])])
    for k, (a, b) in enumerate(zip(*res)):
        close(a, b, 1e-4, 2e-5 * float(b.abs().max()) + 1e-7, "chained vs layer-by-layer after a weight update #%d" % k)


@pytest.mark.parametrize("walker", [False, True])
def test_row_chain_zero_tangent_input_and_a_fresh_input_in_the_middle(walker, monkeypatch):
    """Two corners of mdg_row_chain's contract: a dual chain whose first stage has no tangent rows (in1 = NULL: x_dot = 0, as
    below the first interaction block) -- on the compiled forward chain and on the walker --, and a later stage that takes
    its input from global memory again instead of the previous stage's outputs."""
    from mdgrad_amd import ops
    if walker:
        monkeypatch.setenv("MDG_CHAIN_WALKER", "1")
    else:
        monkeypatch.delenv("MDG_CHAIN_WALKER", raising=False)
    torch.manual_seed(5)
    N, F, A = 333, 128, 64
    rn = lambda *s: torch.randn(*s, device=DEV)
    U1, U2, Wn = rn(A, F) / F ** 0.5, rn(A, A) / A ** 0.5, rn(F, A) / A ** 0.5
    c1, c2, bn = rn(A), rn(A), rn(F)
    m, r, rd = rn(N, F), rn(N, A), rn(N, A)
    ln2 = float(np.log(2.0))
    ch = ops.RowChain(N, True, m.device)
    a = ch.stage(U1, bias=c1, act=True, in0=m, in1=None, want_sig=True)
    b = ch.stage(U2, bias=c2, res0=r, res1=rd)
    c = ch.stage(Wn, bias=bn)
    ch.run()
    z = m @ U1.t() + c1
    t = torch.nn.functional.softplus(z) - ln2
    rr = t @ U2.t() + c2 + r
    _close(a.out0, t, "t"); _close(a.sig, torch.sigmoid(z), "su")
    assert float(a.out1.abs().max()) == 0.0, "zero tangent in, zero tangent row out of the activation stage"
    _close(b.out0, rr, "r'"); _close(b.out1, rd, "rd' = residual only")
    _close(c.out0, rr @ Wn.t() + bn, "h'"); _close(c.out1, rd @ Wn.t(), "hd'")
    # a stage in the middle with its own global input
    y = rn(N, A)
    ch = ops.RowChain(N, False, m.device)
    ch.stage(U1, bias=c1, in0=m, store=False)
    s2 = ch.stage(U2, bias=c2, in0=y)
    s3 = ch.stage(Wn)
    ch.run()
    _close(s2.out0, y @ U2.t() + c2, "second stage reads its own input")
    _close(s3.out0, (y @ U2.t() + c2) @ Wn.t(), "third stage continues from the second")
