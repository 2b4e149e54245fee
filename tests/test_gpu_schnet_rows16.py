"""SchNet.node_rows_bf16 / mdg_cfconv_*_rows16: the convolution kernels gather bf16 MIRRORS of the node matrices
(include/mdgrad_hip.h).  A precision option on top of the bf16 MFMA operands of BASELINE config #5 -- so this file states
what it costs, at three levels:

  1. the kernels do what they say: fed mirrors, they return what the f32-row bf16 kernels return on the ROUNDED rows
     (forward: the same bits; reverse: the contraction over the filters runs in a permuted order, and an intermediate that
     is rounded to bf16 for the next MFMA can land on the other side of a rounding boundary -- so to bf16 rounding of single
     entries, and no further from the f32-MFMA kernels than the bf16 kernels are);
  2. the mirrors are round-to-nearest-even copies (mdg_rows_to_bf16 and the row chain's out0_h / out1_h against torch's cast);
  3. the model: force, d(w.F)/dx, d(w.F)/dtheta and an NH-Verlet trajectory + adjoint with the option on, against oracle/
     (nff/nn/models/schnet.py:113-171 under torchmd/sovlers.py:211-293) with the tolerance written here, next to the
     deviation of the bf16-operand path without the option on the same inputs.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from test_gpu_parity import T, close, mk_system, DEV
from test_gpu_fused_block import _setup

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())


def test_rows_to_bf16_and_chain_mirrors_are_round_to_nearest_even_copies():
    from mdgrad_amd import ops
    torch.manual_seed(5)
    for N, M in ((1000, 128), (37, 64), (4096, 256), (3, 4)):
        x = torch.randn(N, M, device=DEV) * torch.logspace(-6, 6, M, device=DEV)[None, :]
        x[0, 0], x[-1, -1] = 0.0, -0.0
        assert torch.equal(ops.rows_to_bf16(x), x.to(torch.bfloat16)), (N, M)
    # mirrors written by mdg_row_chain: both kernels (the compiled 64 / 128 chain shape and the descriptor walker)
    for N, A, F in ((4096, 64, 128), (333, 48, 96)):
        W1, b1 = torch.randn(A, F, device=DEV) / F ** 0.5, torch.randn(A, device=DEV)
        W2, b2 = torch.randn(A, A, device=DEV) / A ** 0.5, torch.randn(A, device=DEV)
        W3, b3 = torch.randn(F, A, device=DEV) / A ** 0.5, torch.randn(F, device=DEV)
        m, md, r, rd = [torch.randn(N, k, device=DEV) for k in (F, F, A, A)]
        for dual in (True, False):
            ch = ops.RowChain(N, dual, DEV)
            ch.stage(W1, bias=b1, act=True, in0=m, in1=md if dual else None, want_sig=True)
            ch.stage(W2, bias=b2, res0=r, res1=rd if dual else None)
            c = ch.stage(W3, bias=b3, mirror=True)
            ch.run()
            assert c.out0_h.dtype == torch.bfloat16 and torch.equal(c.out0_h, c.out0.to(torch.bfloat16)), (N, dual)
            if dual:
                assert torch.equal(c.out1_h, c.out1.to(torch.bfloat16))
            else:
                assert c.out1_h is None


@pytest.mark.parametrize("G,F,n_side", [(30, 128, 6), (41, 128, 6), (30, 96, 6), (30, 256, 6), (30, 128, 16)])
def test_rows16_kernels_equal_the_bf16_kernels_on_rounded_rows(G, F, n_side):
    """Level 1.  Mirrors of (h, hd, mb, mdb) into the rows16 kernels vs the rounded f32 rows into the bf16 kernels."""
    from mdgrad_amd import ops
    x, topo, net = _setup(G, F, seed=7 * G + F, n_side=n_side)
    N, E = topo.n_atoms, topo.n_edges
    w = torch.randn(N, 3, device=DEV)
    d, uhat, dd, ddel = ops.edge_geom(x, topo, w)
    f16, f32 = ops.FilterNet(*net, bf16=True, rows16=True), ops.FilterNet(*net, bf16=True)
    assert f16.rows16 and not f32.rows16
    rows = [torch.randn(N, F, device=DEV) for _ in range(4)]
    h16, hd16, mb16, mdb16 = [ops.rows_to_bf16(v) for v in rows]
    h, hd, mb, mdb = [v.float() for v in (h16, hd16, mb16, mdb16)]
    for args16, args32, sums in (((h16, None), (h, None), True), ((h16, hd16), (h, hd), True), ((h16, None), (h, None), False)):
        for tangent in (False, True):
            if not tangent and args16[1] is not None:
                continue
            a = ops.cfconv_fwd(f16, d, dd if tangent else None, args16[0], args16[1] if tangent else None, topo, want_sums=sums)
            b = ops.cfconv_fwd(f32, d, dd if tangent else None, args32[0], args32[1] if tangent else None, topo, want_sums=sums)
            for u, v in zip(a, b):
                assert (u is None) == (v is None)
                if u is not None:
                    assert u.dtype == torch.float32 and torch.equal(u, v), "forward sweep over mirrors: the same bits"
    fex = ops.FilterNet(*net)                     # f32 MFMA, exact f32: the yardstick for both bf16 variants
    for with_hd in (True, False):
        out = []
        for fn, (a, b, c, e) in ((f16, (h16, hd16, mb16, mdb16)), (f32, (h, hd, mb, mdb)), (fex, (h, hd, mb, mdb))):
            d_b, dd_b = torch.zeros(E, device=DEV), torch.zeros(E, device=DEV)
            th = ops.cfconv_bwd(fn, d, dd, topo, a, b if with_hd else None, c, e, d_b, dd_b, want_theta=True)
            d_b2, dd_b2 = torch.zeros(E, device=DEV), torch.zeros(E, device=DEV)
            assert ops.cfconv_bwd(fn, d, dd, topo, a, b if with_hd else None, c, e, d_b2, dd_b2) is None
            plain = torch.zeros(E, device=DEV)
            ops.cfconv_bwd(fn, d, None, topo, a, None, None, e, None, plain)
            out.append([d_b, dd_b, d_b2, dd_b2, plain] + list(th))
        for k, (u, v, x) in enumerate(zip(*out)):
            # the same bf16 operands, summed over the filters in another order (f32 accumulation): s_b = W_b W2 differs in
            # the last f32 bits, and a_b -- rounded to bf16 as the operand of g_b = a_b W1 -- flips a rounding now and then
            # (observed: 1.4e-4 of the largest entry on 4096 atoms)
            scale = float(x.abs().max())
            close(u, v, 0, 2e-3 * scale, "reverse sweep over mirrors #%d (hd=%s)" % (k, with_hd))
            e16, e32 = float((u - x).abs().max()), float((v - x).abs().max())
            assert e16 <= 1.5 * e32 + 1e-3 * scale, "reverse sweep #%d: mirrors %.2e, rounded f32 rows %.2e from the f32-MFMA kernels (scale %.2e)" % (k, e16, e32, scale)
    # unsupported widths say so instead of reading garbage
    if F == 128:
        small = _setup(G, 64, seed=1, n_side=n_side)[2]
        assert not ops.FilterNet(*small, bf16=True, rows16=True).rows16
        assert not ops.FilterNet(*net, bf16=False, rows16=True).rows16, "bf16 rows go with the bf16 filter kernels"


def _model(convs=2, A=64, F=128, G=30, R=1, seed=11):
    from mdgrad_amd.interface import GNNPotentials
    from mdgrad_amd.nn import get_model
    g = load_golden("schnet_cg64")
    system = mk_system(g["pos"], g["cell"], mass=g["masses"], numbers=g["numbers"])
    pos = g["pos"]
    if R > 1:
        system = system.replicate(R)
        rng = np.random.default_rng(17)
        pos = np.concatenate([np.mod(g["pos"] + rng.normal(0, 0.03, g["pos"].shape), g["cell"]) for _ in range(R)]).astype(np.float32)
        system.set_positions(pos)
    torch.manual_seed(seed)
    net = get_model({"n_atom_basis": A, "n_filters": F, "n_gaussians": G, "n_convolutions": convs, "cutoff": 6.0})
    gnn = GNNPotentials(system, net, cutoff=6.0)
    return g, system, net, gnn, pos


@pytest.mark.parametrize("convs,R", [(2, 1), (3, 1), (1, 1), (2, 40)])
def test_force_and_force_vjp_with_bf16_node_rows_vs_f32_path(convs, R):
    """Level 3a.  analytic.force / force_vjp (energy, F, d(w.F)/dx, d(w.F)/dtheta) on the row-chain path: all-f32 (pinned to
    oracle/ at 2e-5 in tests/test_gpu_schnet.py), bf16 filter operands, and bf16 operands + bf16 node rows.  The option's
    deviation from f32 stays within TWICE the stated tolerance of the bf16-operand path (tests/test_gpu_config5.py: 5e-3 of the
    largest entry for forces / vjps, 1e-2 for the parameter gradient), and is reported next to the bf16-operand path's."""
    from mdgrad_amd.nn import analytic
    g, system, net, gnn, pos = _model(convs=convs, R=R)
    q = T(pos, DEV)
    gnn._reset_topology(q)
    w = T(np.random.default_rng(3).normal(0, 1, pos.shape).astype(np.float32), DEV)
    res = {}
    for mode, (fb, rb) in (("f32", (False, False)), ("bf16", (True, False)), ("rows16", (True, True))):
        net.filter_bf16, net.node_rows_bf16 = fb, rb
        assert analytic.chain_ok(net)
        U, F_ = analytic.force(net, gnn._z(), q, gnn.inputs["_topo"])
        U2, F2, dq, gth = analytic.force_vjp(net, gnn._z(), q, w, gnn.inputs["_topo"])
        _, F3, dq3, none = analytic.force_vjp(net, gnn._z(), q, w, gnn.inputs["_topo"], want_theta=False, want_energy=False)
        assert none is None
        res[mode] = dict(U=U.reshape(1), F=F_, F2=F2, dq=dq, F3=F3, dq3=dq3, th=torch.cat([t.reshape(-1) for t in gth]))
    tol = dict(U=None, F=1e-2, F2=1e-2, dq=1e-2, F3=1e-2, dq3=1e-2, th=2e-2)
    for k, t in tol.items():
        e16, er = _rel(res["bf16"][k], res["f32"][k]), _rel(res["rows16"][k], res["f32"][k])
        print("DEV convs=%d R=%d %-3s rel. to largest entry: bf16 operands %.2e   + bf16 node rows %.2e" % (convs, R, k, e16, er))
        if t is None:        # the energy is a sum over atoms that cancels (3e-2 of |U| with bf16 operands alone): relative to that
            t = 1.5 * e16 + 1e-3
        assert er <= t, "%s: bf16 node rows deviate %.2e from the f32 path (allowed %.1e; bf16 operands alone: %.2e)" % (k, er, t, e16)
    # the option is really on: the results differ from the bf16-operand path, and it is bitwise reproducible
    assert not torch.equal(res["rows16"]["F"], res["bf16"]["F"])
    net.filter_bf16, net.node_rows_bf16 = True, True
    again = analytic.force_vjp(net, gnn._z(), q, w, gnn.inputs["_topo"])
    assert torch.equal(again[2], res["rows16"]["dq"]) and torch.equal(torch.cat([t.reshape(-1) for t in again[3]]), res["rows16"]["th"])


def test_trajectory_and_adjoint_with_bf16_node_rows_vs_oracle():
    """Level 3b.  bench.py's SchNet workload at 216 and 512 beads (same builder, same model, same prior) with the option on:
    4 NH-Verlet steps + RDF loss + analytic adjoint against oracle/ on the same inputs.  Tolerances: those of the bf16-operand
    path (tests/test_gpu_secondary_pins.py::_tols(True)) for the trajectory and g(r), twice those for the adjoints and the
    parameter gradient."""
    import bench
    import oracle as O
    from mdgrad_amd import units
    from test_gpu_secondary_pins import _run_schnet_workload, _rdf_plus, _check_replica, _check_theta, _tols
    for size in (3, 4):
        wl = bench.build_schnet_workload(DEV, 1, True, 50 + size, size=size, rows16=True)
        assert wl["net"].node_rows_bf16 and wl["net"].filter_bf16
        sd = {k: v.detach().clone().cpu() for k, v in wl["net"].state_dict().items()}
        pos = wl["system"].get_positions().astype(np.float32)
        vel = wl["system"].get_velocities().astype(np.float32)
        t = torch.Tensor([units.fs * i for i in range(5)])
        out = _run_schnet_workload(wl, t, 2)
        cellt = torch.tensor([wl["L"]] * 3, dtype=torch.float32)
        traj, lam, gth = bench.schnet_oracle_replica(wl, sd, pos, vel, t, _rdf_plus(cellt, 2, wl["N"]))
        g = O.rdf_oracle(traj[1][::2], cellt, 60, (2.0, 6.0))[2]
        tol = dict(_tols(True))
        tol["adj"], tol["th"], tol["cos"] = (0.0, 4e-2), (0.0, 1e-2), 0.9995
        _check_replica(out, 0, (traj, lam, gth, g), tol, "%d beads, bf16 operands + bf16 node rows" % wl["N"])
        _check_theta(out["flat"], gth, tol, "%d beads, bf16 operands + bf16 node rows" % wl["N"])
