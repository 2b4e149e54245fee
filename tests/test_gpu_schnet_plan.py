"""`mdg_schnet_force` / `mdg_schnet_force_vjp` (csrc/schnet_eval.hip: ONE C-ABI call per SchNet evaluation, the launches
enqueued by a C++ loop) against the launch-by-launch sequence of mdgrad_amd/nn/analytic.py it replaces -- which is pinned to
the reference's own outputs (goldens G8 / G9 / G14 / G15 / G17, tests/test_gpu_parity.py, test_gpu_pins.py).  Same kernels,
same operands, same order: the comparison is BITWISE, for every configuration the plan takes."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from test_gpu_parity import T, mk_system, DEV

pytestmark = pytest.mark.gpu


def _net(A, F, G, n_conv, seed, bf16=False, rows16=False):
    from mdgrad_amd.nn import get_model
    torch.manual_seed(seed)
    net = get_model({"n_atom_basis": A, "n_filters": F, "n_gaussians": G, "n_convolutions": n_conv, "cutoff": 5.0}).to(DEV)
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.05)            # (biases are zero-initialised: give every bias path something to carry)
    net.filter_bf16, net.node_rows_bf16 = bool(bf16), bool(rows16)
    return net


def _system(n_side, seed, cutoff=5.0, kind="graph"):
    from mdgrad_amd import ops, _lib
    rng = np.random.default_rng(seed)
    L = 2.9 * n_side
    g = np.stack(np.meshgrid(*[np.arange(n_side)] * 3, indexing="ij"), -1).reshape(-1, 3) * (L / n_side)
    pos = np.mod(g + rng.normal(0, 0.35, g.shape), L).astype(np.float32)
    x = T(pos, DEV)
    cell = _lib.make_cell(np.array([L, L, L], dtype=np.float32))
    ell = ops.build_ell(x, cell, cutoff)
    if kind == "graph":
        topo = ops.GraphTopo(ell)
    else:                                                         # fixed capacity, padded rows (what graph replay / eager_static use)
        need = torch.zeros(2, dtype=torch.int32, device=DEV)
        cap = (int(ops.GraphTopo(ell).n_edges * 1.3) + 1023) // 1024 * 1024
        ell2 = ops.build_ell(x, cell, cutoff, max_nbr=(int(ell.cnt.max()) * 5 // 4 + 15) // 8 * 8, need=need)
        topo = ops.StaticTopo(ell2, cap, need)
    z = torch.full((len(pos),), 8, dtype=torch.long, device=DEV)
    z[::3] = 1
    return x, z, topo


def _both(net, fn):
    """fn() with the plan and with the launch-by-launch path."""
    net.eval_plan = True
    a = fn()
    net.eval_plan = False
    b = fn()
    net.eval_plan = True
    return a, b


def _same(a, b, what):
    if a is None or b is None:
        assert a is None and b is None, what
        return
    assert a.shape == b.shape and torch.equal(a, b), "%s: plan differs from the launch-by-launch path (max |d| %.3e)" % (
        what, float((a - b).abs().max()))


@pytest.mark.parametrize("A,F,G,n_conv,mode", [(64, 128, 30, 2, "f32"), (64, 128, 30, 2, "bf16"), (64, 128, 30, 2, "rows16"),
                                               (128, 128, 32, 3, "f32"), (128, 128, 32, 3, "bf16"), (32, 64, 25, 1, "f32"),
                                               (64, 256, 41, 2, "bf16"), (48, 48, 16, 2, "f32"), (256, 256, 41, 2, "rows16")])
@pytest.mark.parametrize("topo_kind", ["graph", "static"])
def test_one_call_evaluation_is_bitwise_the_launch_by_launch_sequence(A, F, G, n_conv, mode, topo_kind):
    from mdgrad_amd import ops
    from mdgrad_amd.nn import analytic, plan
    net = _net(A, F, G, n_conv, seed=A + F + G, bf16=mode != "f32", rows16=mode == "rows16")
    x, z, topo = _system(6, seed=G, kind=topo_kind)
    assert analytic.fused_ok(net) and analytic.chain_ok(net)
    pl, rows = analytic._plan_inputs(net, z)
    assert pl is not None, "this configuration must take the plan"
    w = torch.randn_like(x)
    # first order
    (Ua, Fa), (Ub, Fb) = _both(net, lambda: analytic.force(net, z, x, topo, want_energy=True))
    _same(Fa, Fb, "force")
    assert abs(float(Ua) - float(Ub)) <= 1e-5 * abs(float(Ub)) + 1e-5, "energy"       # (same column sums, dot product in another order)
    (_, Fa2), (_, Fb2) = _both(net, lambda: analytic.force(net, z, x, topo, want_energy=False))
    _same(Fa2, Fb, "force without the energy"), _same(Fb2, Fb, "force without the energy (launch by launch)")
    # second order, with and without parameter gradients
    for theta in (True, False):
        (Ua, Fa, da, ga), (Ub, Fb, db, gb) = _both(net, lambda: analytic.force_vjp(net, z, x, w, topo, want_theta=theta, want_energy=True))
        _same(Fa, Fb, "force (vjp, theta=%s)" % theta)
        _same(da, db, "d(w.F)/dx (theta=%s)" % theta)
        assert (ga is None) == (gb is None) == (not theta)
        if theta:
            assert len(ga) == len(gb) == len(list(net.parameters()))
            for p, u, v in zip(net.state_dict(), ga, gb):
                _same(u, v, "d(w.F)/dtheta")
            assert any(float(u.abs().max()) > 0 for u in ga)
    # ... into a caller's accumulator with a device-side interval weight (the adjoint sweep's form, sovlers.py:160)
    t = torch.tensor([0.0, 0.5, 1.25], device=DEV)
    idx = torch.tensor([2], device=DEV)
    outs = []
    for use_plan in (True, False):
        net.eval_plan = use_plan
        acc = ops.ThetaAccum(net.parameters(), t=t, idx=idx)
        acc.flat.fill_(0.25)
        r = analytic.force_vjp(net, z, x, w, topo, want_theta=True, want_energy=False, accum=acc)
        assert r[3] is None and r[0] is None
        outs.append((r[1], r[2], acc.flat.clone()))
    net.eval_plan = True
    for u, v, nm in zip(outs[0], outs[1], ("force", "dwf", "accumulated flat gradient")):
        _same(u, v, nm + " (accumulator)")


def test_plan_follows_replaced_weights_and_is_refused_for_a_trainable_basis():
    from mdgrad_amd.nn import analytic, get_model
    net = _net(64, 128, 30, 2, seed=5)
    x, z, topo = _system(5, seed=2)
    pl, _ = analytic._plan_inputs(net, z)
    F0 = analytic.force(net, z, x, topo, want_energy=False)[1]
    with torch.no_grad():                                          # in-place update (an optimizer step): same plan, new values
        net.convolutions[0].moduledict["update_function"][0].weight.mul_(1.5)
        net.atom_embed.weight.add_(0.01)
    F1 = analytic.force(net, z, x, topo, want_energy=False)[1]
    assert analytic._plan_inputs(net, z)[0] is pl and not torch.equal(F0, F1)
    net.eval_plan = False
    assert torch.equal(analytic.force(net, z, x, topo, want_energy=False)[1], F1)
    net.eval_plan = True
    lin = net.convolutions[1].moduledict["message_node_filter"]    # a replaced Parameter: the plan is rebuilt
    lin.weight = torch.nn.Parameter(lin.weight.detach() * 0.5)
    F2 = analytic.force(net, z, x, topo, want_energy=False)[1]
    assert analytic._plan_inputs(net, z)[0] is not pl
    net.eval_plan = False
    assert torch.equal(analytic.force(net, z, x, topo, want_energy=False)[1], F2)
    torch.manual_seed(0)
    tr = get_model({"n_atom_basis": 64, "n_filters": 128, "n_gaussians": 30, "n_convolutions": 2, "cutoff": 5.0,
                    "trainable_gauss": True}).to(DEV)
    assert analytic._plan_inputs(tr, z)[0] is None, "the basis gradients are chained through torch ops: launch-by-launch path"
    assert torch.isfinite(analytic.force_vjp(tr, z, x, torch.randn_like(x), topo)[2]).all()


def test_plan_under_graph_replay_and_stored_lists_equals_the_launch_by_launch_trajectory():
    """A whole trajectory + adjoint through HIP-graph replay (fixed-capacity Verlet lists, three evaluations per captured
    step) with the plan and without: bit-identical frames, adjoints and parameter gradients; and against golden G9 (the
    reference's own GNN + prior trajectory) within its tolerance."""
    from mdgrad_amd import graphs
    from mdgrad_amd.sovlers import odeint_adjoint
    from test_gpu_parity import close
    from test_gpu_schnet import _gnn_integrator
    g = load_golden("gnn_traj")
    res = []
    for use_plan in (True, False):
        system = mk_system(g["pos"], g["cell"], g["vel"], g["masses"], g["numbers"])
        integ = _gnn_integrator(g, system)
        integ.model.models["gnn"].gnn.eval_plan = use_plan
        assert graphs.enabled(integ)
        y0 = [s.clone().requires_grad_(True) for s in integ.get_inital_states(wrap=True)]
        t = torch.Tensor([float(g["dt"]) * i for i in range(11)]).to(DEV)
        v_t, q_t, pv_t = odeint_adjoint(integ, tuple(y0), t, method="NH_verlet")
        (q_t[::2].pow(2).mean() + v_t[-1].pow(2).mean()).backward()
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in integ.parameters()])
        res.append((q_t.detach(), v_t.detach(), y0[1].grad, y0[0].grad, flat))
    for a, b, nm in zip(res[0], res[1], ("q_t", "v_t", "adj q0", "adj v0", "dL/dtheta")):
        _same(a, b, nm + " (trajectory)")
    close(res[0][0], g["q_t"], 1e-4, 1e-4 * max(1e-3, np.abs(g["q_t"]).max()), "q_t vs the reference (golden G9)")
