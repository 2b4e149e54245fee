"""torch.ops.mdgrad.* (csrc_torch/mdgrad_torch.cpp, the TORCH_LIBRARY layer of SURVEY 8b) against the ctypes bindings of
the same C entry points: identical kernels, so identical bits -- for the ops the Python wrappers route through it
(SchNet interaction block) and for the ones a caller of the reference's op list would use directly (neighbour list,
pair force / Hessian-vector product, trajectory + adjoint, RDF)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from test_gpu_parity import T, mk_system, lj_setup, liquid, DEV

pytestmark = pytest.mark.gpu


def _both(fn):
    """fn() with the op library and with the ctypes path."""
    from mdgrad_amd import _torch_ops
    assert _torch_ops.get() is not None, "libmdgrad_torch.so was not built (python -c 'import __graft_entry__ as g; g.build()')"
    a = fn()
    saved = dict(_torch_ops._state)
    _torch_ops._state.update(tried=True, ns=None)
    try:
        b = fn()
    finally:
        _torch_ops._state.update(saved)
    return a, b


def _same(a, b, what):
    if a is None or b is None:
        assert a is None and b is None, what
        return
    if isinstance(a, (tuple, list)):
        assert len(a) == len(b), what
        for k, (x, y) in enumerate(zip(a, b)):
            _same(x, y, "%s[%d]" % (what, k))
        return
    assert a.shape == b.shape and torch.equal(a, b), what


def test_schnet_block_ops_match_ctypes_bitwise():
    from mdgrad_amd import ops, _lib
    pos, cell = liquid(8, seed=3)
    x = T(pos, DEV)
    ell = ops.build_ell(x, _lib.make_cell(np.asarray(cell, dtype=np.float32)), 2.2)
    topo = ops.GraphTopo(ell)
    g = torch.Generator(device="cpu").manual_seed(0)
    G_, F_, A_ = 32, 64, 64
    rnd = lambda *s: torch.randn(*s, generator=g).to(DEV)
    fnet = ops.FilterNet(torch.linspace(0, 2.2, G_).to(DEV), torch.full((G_,), -10.0).to(DEV), rnd(G_, G_) * 0.2, rnd(G_) * 0.1,
                         rnd(F_, G_) * 0.2, rnd(F_) * 0.1)
    w = rnd(x.shape[0], 3)
    h, hd = rnd(x.shape[0], F_), rnd(x.shape[0], F_)
    geo = _both(lambda: ops.edge_geom(x, topo, w))
    _same(*geo, "edge_geom")
    d, uhat, dd, ddel = geo[0]
    fwd = _both(lambda: ops.cfconv_fwd(fnet, d, dd, h, hd, topo, want_sums=True))
    _same(*fwd, "cfconv_fwd")
    _same(*_both(lambda: ops.cfconv_fwd(fnet, d, None, h, None, topo)), "cfconv_fwd primal")
    mb, mdb = rnd(x.shape[0], F_), rnd(x.shape[0], F_)

    def bwd():
        d_b, dd_b = torch.zeros_like(d), torch.zeros_like(d)
        th = ops.cfconv_bwd(fnet, d, dd, topo, h, hd, mb, mdb, d_b, dd_b, want_theta=True)
        return (d_b, dd_b) + tuple(th)
    _same(*_both(bwd), "cfconv_bwd")
    _same(*_both(lambda: ops.edge_geom_bwd(rnd(d.shape[0]).mul(0) + d, dd, d, dd, uhat, ddel, topo)), "edge_geom_bwd")
    W = rnd(A_, F_) * 0.1
    _same(*_both(lambda: ops.dense(W, h, bias=rnd(A_).mul(0) + 0.1, act=True, x1=hd, want_sig=True)), "dense")
    _same(*_both(lambda: ops.dense(W, rnd(x.shape[0], A_).mul(0) + 0.3, trans=True)), "dense trans")
    _same(*_both(lambda: ops.ssp_dual_bwd_t(h.sigmoid(), hd, mb, mdb)), "ssp_dual_bwd_t")
    _same(*_both(lambda: ops._atb(h, hd)), "atb")


def test_reference_op_list_through_torch_ops():
    """nbr_build -> pair_force / pair_hvp -> nhc_vv_forward / nhc_vv_adjoint -> rdf_fwd / rdf_bwd called directly as
    torch.ops.mdgrad.*, against the Python layer (which is pinned to the goldens elsewhere)."""
    from mdgrad_amd import ops, _torch_ops
    from mdgrad_amd.observable import rdf
    ns = _torch_ops.get()
    assert ns is not None
    g = load_golden("nhc_traj_lj")
    system, mdl, integ = lj_setup(g)
    spec = integ.fused_spec("NH_verlet")
    cell = _torch_ops.cell_args(spec.cell_struct)
    x = T(g["pos"], DEV)
    # neighbour list
    ell = ops.build_ell(x, spec.cell_struct, 2.5)
    col, shift, cnt, ovf = ns.nbr_build(x, cell, 2.5, None, ell.max_nbr, 0, False)
    assert int(ovf) == 0 and torch.equal(cnt, ell.cnt)
    k = torch.arange(ell.max_nbr, device=DEV)[None] < cnt[:, None]
    assert torch.equal(col[k], ell.col[k]) and torch.equal(shift[k], ell.shift[k])
    # pair force / Hessian-vector product
    theta = spec.flat_params().detach()
    term = spec.terms.t[0]
    ti, tf = [term.kind, term.p, term.q, term.theta_off, term.n_theta], [term.c, term.a, term.phi, term.cutoff]
    U, dU, gth = ns.pair_force(x, cell, col, shift, cnt, ti, tf, None, theta)
    o = ops.pair_eval(ell, x, term, theta, energy=True, grad=True)
    assert torch.equal(U.reshape(()), o["energy"].reshape(())) and torch.equal(dU, o["grad"]) and torch.equal(gth[:2], o["gtheta"][:2])
    w = torch.randn_like(x)
    hw, gw = ns.pair_hvp(x, cell, col, shift, cnt, ti, tf, None, theta, w)
    o2 = ops.pair_eval(ell, x, term, theta, w=w)
    assert torch.equal(hw, o2["hw"]) and torch.equal(gw[:2], o2["gtheta_w"][:2])
    # trajectory + adjoint
    R, nT = 2, 8
    v0, q0 = T(np.stack([g["vel"]] * R), DEV), T(np.stack([g["pos"]] * R), DEV)
    pv0 = torch.zeros(R, 5, device=DEV)
    t = torch.Tensor([0.005 * i for i in range(nT)]).to(DEV)
    prm = spec.params(R, nT)
    iprm = [R, 108, nT, 5, 0, 0]
    fprm = [float(prm.T), float(prm.n_dof)] + [float(prm.Q[c]) for c in range(5)]
    v_t, q_t, pv_t, bad = ns.nhc_vv_forward(v0, q0, pv0, spec.mass, t, theta, iprm, fprm, cell, ti, tf, 2)
    ref = ops.FusedTrajFn.apply(v0, q0, pv0, t, theta, spec)
    assert torch.equal(v_t, ref[0]) and torch.equal(q_t, ref[1]) and torch.equal(pv_t, ref[2]) and int(bad.sum()) == 0
    gq = torch.randn_like(q_t)
    av, aq, ap, ath = ns.nhc_vv_adjoint(v_t, q_t, pv_t, None, gq, None, spec.mass, t, theta, iprm, fprm, cell, ti, tf, 2)
    v0r, q0r, p0r = v0.clone().requires_grad_(True), q0.clone().requires_grad_(True), pv0.clone().requires_grad_(True)
    th_r = theta.clone().requires_grad_(True)
    out = ops.FusedTrajFn.apply(v0r, q0r, p0r, t, th_r, spec)
    (out[1] * gq).sum().backward()
    assert torch.equal(av, v0r.grad) and torch.equal(aq, q0r.grad) and torch.equal(ap, p0r.grad)
    assert torch.equal(ath.sum(0), th_r.grad)
    # rdf
    obs = rdf(system, nbins=100, r_range=(0.75, 2.5))
    xyz = q_t.reshape(-1, 108, 3)
    raw = ns.rdf_fwd(xyz, cell, obs.cutoff_boundary, None, obs.offsets, obs.spacing, obs.coeff)
    xr = xyz.clone().requires_grad_(True)
    raw_ref = ops.RdfRawFn.apply(xr, obs.offsets, obs.coeff, obs.cutoff_boundary, obs._cell_struct, None, obs.spacing)
    assert torch.equal(raw, raw_ref.detach())
    g_raw = torch.randn_like(raw)
    (raw_ref * g_raw).sum().backward()
    assert torch.equal(ns.rdf_bwd(xyz, cell, obs.cutoff_boundary, None, obs.offsets, obs.spacing, obs.coeff, g_raw), xr.grad)


def test_torch_ops_reject_bad_input():
    from mdgrad_amd import _torch_ops
    ns = _torch_ops.get()
    with pytest.raises(RuntimeError):
        ns.atb(torch.zeros(8, 4), torch.zeros(8, 4))                       # CPU tensors: no such backend
    with pytest.raises(RuntimeError, match="float32"):
        ns.atb(torch.zeros(8, 4, device=DEV, dtype=torch.float64), torch.zeros(8, 4, device=DEV))
    with pytest.raises(RuntimeError, match="contiguous"):
        ns.atb(torch.zeros(4, 8, device=DEV).t(), torch.zeros(8, 4, device=DEV))
