"""GPU parity of the SchNet / GNNPotentials path against the reference
goldens G8/G9 and against plain torch index ops for the graph kernels."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from test_gpu_parity import T, close, mk_system, DEV

pytestmark = pytest.mark.gpu


def sd_of(g):
    sd = {k[4:]: T(v) for k, v in g.items() if k.startswith("sd__")}
    if "embed_row8" in g:              # (wide goldens keep the one embedding row in use: Z = 8; rows never read are zeros)
        emb = torch.zeros(100, int(g["n_atom_basis"]))
        emb[8] = T(g["embed_row8"])
        sd = dict([("atom_embed.weight", emb)] + list(sd.items()))
    return sd


def params_of(g):
    return {"n_atom_basis": int(g["n_atom_basis"]), "n_filters": int(g["n_filters"]),
            "n_gaussians": int(g["n_gaussians"]), "n_convolutions": int(g["n_convolutions"]),
            "cutoff": float(g["cutoff"])}


def test_graph_ops_match_torch_index_ops_to_second_order():
    from mdgrad_amd import ops, _lib
    g = load_golden("schnet_water192")
    x = T(g["pos"], DEV)
    ell = ops.build_ell(x, _lib.make_cell(g["cell"]), 5.0)
    topo = ops.GraphTopo(ell)
    assert np.array_equal(topo.nbr.cpu().numpy(), g["nbr"])
    a0, a1 = topo.nbr[:, 0], topo.nbr[:, 1]
    torch.manual_seed(0)
    F = 40
    h = torch.randn(192, F, device=DEV, requires_grad=True)
    W = torch.randn(topo.n_edges, F, device=DEV, requires_grad=True)
    xx = x.clone().requires_grad_(True)

    def ref(h, W, xx):
        m = torch.zeros_like(h).index_add(0, a1, h[a0] * W).index_add(0, a0, h[a1] * W)
        d = (xx[a0] - xx[a1]).pow(2).sum(1)
        return m, d

    def hip(h, W, xx):
        return ops.CfconvAggFn.apply(h, W, topo), ops.EdgeDiffFn.apply(xx, topo).pow(2).sum(1)

    outs = []
    for fn in (ref, hip):
        m, d = fn(h, W, xx)
        y = (m.tanh() * torch.linspace(0.5, 1.5, F, device=DEV)).sum() + (d.sqrt() * W[:, 0]).sum()
        g1 = torch.autograd.grad(y, [h, W, xx], create_graph=True)
        z = sum((gi * gi.detach().cos()).sum() for gi in g1)
        g2 = torch.autograd.grad(z, [h, W, xx])
        outs.append([m, d] + list(g1) + list(g2))
    for a, b in zip(*outs):
        close(a, b, 2e-4, 2e-5 * float(b.abs().max()) + 1e-6, "graph op")


@pytest.mark.parametrize("G,F,E", [(16, 48, 1000), (30, 128, 777), (64, 256, 130), (12, 20, 64), (33, 130, 5)])
def test_mfma_filter_kernel_vs_torch_to_second_order(G, F, E):
    from mdgrad_amd import ops
    torch.manual_seed(G * F)
    d = (torch.rand(E, device=DEV) * 5.0).requires_grad_(True)
    mu = torch.linspace(0, 5.0, G, device=DEV).requires_grad_(True)
    width = torch.full((G,), float(5.0 / (G - 1)), device=DEV).requires_grad_(True)
    W1 = (torch.randn(G, G, device=DEV) / G ** 0.5).requires_grad_(True)
    b1 = (torch.randn(G, device=DEV) * 0.1).requires_grad_(True)
    W2 = (torch.randn(F, G, device=DEV) / G ** 0.5).requires_grad_(True)
    b2 = (torch.randn(F, device=DEV) * 0.1).requires_grad_(True)
    ins = [d, mu, width, W1, b1, W2, b2]
    probe = torch.randn(E, F, device=DEV)
    outs = []
    for fn in (ops.filter_reference, ops.CfconvFilterFn.apply):
        W = fn(*ins)
        g1 = torch.autograd.grad((W * probe).sum(), ins, create_graph=True)
        z = sum((gi * gi.detach().sin()).sum() for gi in g1)
        g2 = torch.autograd.grad(z, ins, allow_unused=True)
        outs.append([W] + list(g1) + [x if x is not None else torch.zeros(1, device=DEV) for x in g2])
    for k, (a, b) in enumerate(zip(*outs)):
        close(b, a, 2e-4, 2e-5 * float(a.abs().max()) + 1e-6, "filter output/derivative #%d" % k)


@pytest.mark.parametrize("name", ["schnet_cg64", "schnet_water192", "schnet_cg64_wide", "schnet_cg64_a256"])
def test_schnet_energy_force_vjp_golden(name):
    from mdgrad_amd.interface import GNNPotentials
    from mdgrad_amd.nn import get_model
    g = load_golden(name)
    system = mk_system(g["pos"], g["cell"], mass=g["masses"], numbers=g["numbers"])
    net = get_model(params_of(g))
    net.load_state_dict(sd_of(g))
    gnn = GNNPotentials(system, net, cutoff=float(g["cutoff"]))
    q = T(g["pos"], DEV).requires_grad_(True)
    gnn._reset_topology(q.detach())
    assert np.array_equal(gnn.inputs["nbr_list"].cpu().numpy(), g["nbr"])
    assert np.array_equal(gnn.inputs["offsets"].cpu().numpy(), g["offsets"])
    U = gnn(q)
    assert U.shape == (1, 1)
    close(U.reshape(-1), g["U"], 1e-5, 1e-5, "U")
    (gq,) = torch.autograd.grad(U.sum(), q, create_graph=True)
    close(-gq, g["F"], 1e-4, 1e-5 * np.abs(g["F"]).max(), "F")
    if "w" in g:
        plist = list(net.parameters())
        grads = torch.autograd.grad((T(g["w"], DEV) * -gq).sum(), [q] + plist, allow_unused=True)
        close(grads[0], g["dwF_dq"], 1e-3, 1e-4 * np.abs(g["dwF_dq"]).max(), "d(w.F)/dq")
        flat = torch.cat([(x if x is not None else torch.zeros_like(p)).reshape(-1) for x, p in zip(grads[1:], plist)])
        close(flat, g["dwF_dtheta"], 1e-3, 1e-4 * np.abs(g["dwF_dtheta"]).max(), "d(w.F)/dtheta")


def test_gnn_stack_trajectory_adjoint_golden():
    """BASELINE config #3 shape: Stack(SchNet GNN + ExcludedVolume prior), NHC, adjoint of an RDF loss."""
    from mdgrad_amd import potentials as P
    from mdgrad_amd.interface import GNNPotentials, PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    from mdgrad_amd.nn import get_model
    from mdgrad_amd.observable import rdf
    from mdgrad_amd.sovlers import odeint_adjoint
    g = load_golden("gnn_traj")
    system = mk_system(g["pos"], g["cell"], g["vel"], g["masses"], g["numbers"])
    net = get_model(params_of(g))
    net.load_state_dict(sd_of(g))
    gnn = GNNPotentials(system, net, cutoff=float(g["cutoff"]))
    prior = PairPotentials(system, P.ExcludedVolume(float(g["prior_sigma"]), float(g["prior_epsilon"]), 12),
                           cutoff=float(g["cutoff"]))
    integ = NoseHooverChain(Stack({"gnn": gnn, "prior": prior}), system, T=float(g["T"]),
                            num_chains=int(g["chains"]), Q=float(g["Q"]), adjoint=True).to(DEV)
    assert integ.fused_spec("NH_verlet") is None
    assert [n for n, _ in integ.named_parameters()] == [str(x) for x in g["param_names"]]
    y0 = [s.clone().requires_grad_(True) for s in integ.get_inital_states(wrap=True)]
    t = torch.Tensor([float(g["dt"]) * i for i in range(11)]).to(DEV)
    v_t, q_t, pv_t = odeint_adjoint(integ, tuple(y0), t, method="NH_verlet")
    for x, k in zip((v_t, q_t, pv_t), ["v_t", "q_t", "pv_t"]):
        close(x, g[k], 1e-4, 1e-4 * max(1e-3, np.abs(g[k]).max()), k)
    _, _, gr = rdf(system, nbins=40, r_range=(2.0, 5.5))(q_t[::2])
    close(gr, g["g"], 1e-3, 2e-4, "g")
    loss = gr.pow(2).mean() + q_t[-1].pow(2).mean() * 1e-3
    close(loss.reshape(1), g["loss"], 1e-4, 1e-6, "loss")
    loss.backward()
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in integ.parameters()])
    close(flat, g["grad_flat"], 5e-3, 2e-4 * np.abs(g["grad_flat"]).max(), "dL/dtheta (14339 params)")
    close(y0[1].grad, g["grad_q0"], 5e-3, 2e-3 * np.abs(g["grad_q0"]).max(), "grad_q0")
    close(y0[0].grad, g["grad_v0"], 5e-3, 2e-3 * np.abs(g["grad_v0"]).max(), "grad_v0")


def _gnn_integrator(g, system, seed_net=None):
    from mdgrad_amd import potentials as P
    from mdgrad_amd.interface import GNNPotentials, PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    from mdgrad_amd.nn import get_model
    net = get_model(params_of(g))
    net.load_state_dict(sd_of(g))
    gnn = GNNPotentials(system, net, cutoff=float(g["cutoff"]))
    prior_model = P.ExcludedVolume(float(g["prior_sigma"]), float(g["prior_epsilon"]), 12)
    prior = PairPotentials(system, prior_model, cutoff=float(g["cutoff"]))
    integ = NoseHooverChain(Stack({"gnn": gnn, "prior": prior}), system, T=float(g["T"]),
                            num_chains=int(g["chains"]), Q=float(g["Q"]), adjoint=True).to(DEV)
    return integ


def test_stacked_replicas_generic_gnn_equals_separate_runs():
    """System.replicate: R replicas of the CG-water box in ONE generic SchNet trajectory (per-replica
    thermostats, group-restricted neighbour lists) == R separate runs; gradients add up."""
    from mdgrad_amd.observable import rdf
    from mdgrad_amd.sovlers import odeint_adjoint
    g = load_golden("gnn_traj")
    R, N = 3, 64
    rng = np.random.default_rng(5)
    pos = np.stack([np.mod(g["pos"] + rng.normal(0, 0.05, g["pos"].shape), g["cell"]) for _ in range(R)])
    vel = np.stack([g["vel"] * (1 + 0.2 * r) for r in range(R)])
    t = torch.Tensor([float(g["dt"]) * i for i in range(5)]).to(DEV)

    def loss_of(q_t, obs):
        return obs(q_t)[2].pow(2).mean() + q_t[-1].pow(2).mean() * 1e-3

    base = mk_system(g["pos"], g["cell"], g["vel"], g["masses"], g["numbers"])
    stacked = base.replicate(R)
    stacked.set_positions(pos.reshape(-1, 3))
    stacked.set_velocities(vel.reshape(-1, 3))
    integ = _gnn_integrator(g, stacked)
    assert integ.n_rep == R and integ.n_group == N
    y0 = tuple(integ.get_inital_states(wrap=True))
    assert y0[2].shape == (R, 5)
    v_t, q_t, pv_t = odeint_adjoint(integ, y0, t, method="NH_verlet")
    assert q_t.shape == (5, R * N, 3) and pv_t.shape == (5, R, 5)
    obs_s = rdf(stacked, nbins=40, r_range=(2.0, 5.5))
    (loss_of(q_t, obs_s) * 1.0).backward()
    g_stack = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in integ.parameters()])

    # the stacked rdf pools R*T frames: reproduce with separate runs by pooling their frames
    qs, integs = [], []
    for r in range(R):
        sysr = mk_system(pos[r], g["cell"], vel[r], g["masses"], g["numbers"])
        ir = _gnn_integrator(g, sysr)
        out = odeint_adjoint(ir, tuple(ir.get_inital_states(wrap=True)), t, method="NH_verlet")
        close(q_t.reshape(5, R, N, 3)[:, r], out[1], 1e-4, 2e-5 * float(out[1].abs().max()), "q_t replica %d" % r)
        close(pv_t[:, r], out[2], 1e-3, 1e-5, "pv_t replica %d" % r)
        qs.append(out[1])
        integs.append(ir)
    obs_1 = rdf(base, nbins=40, r_range=(2.0, 5.5))
    pooled = torch.stack(qs, 1)                                   # [T, R, N, 3]
    l2 = obs_1(pooled)[2].pow(2).mean() + sum(q[-1].pow(2).sum() for q in qs) / (R * N * 3) * 1e-3
    l2.backward()
    g_sep = sum(torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                           for p in ir.parameters()]) for ir in integs)
    close(g_stack, g_sep, 5e-3, 5e-4 * float(g_sep.abs().max()), "stacked vs separate dL/dtheta")


@pytest.mark.parametrize("E,M,N", [(100000, 128, 30), (7, 5, 3), (4099, 48, 16), (250000, 16, 130), (0, 8, 8)])
def test_tall_skinny_atb_kernel_and_closure(E, M, N):
    from mdgrad_amd import ops
    torch.manual_seed(E + M)
    A = torch.randn(E, M, device=DEV, requires_grad=True)
    B = torch.randn(E, N, device=DEV, requires_grad=True)
    ref = A.t().matmul(B)
    got = ops.AtBFn.apply(A, B)
    close(got, ref, 1e-4, 1e-4 * (float(ref.abs().max()) + 1e-3), "A^T B")
    assert torch.equal(got, ops.AtBFn.apply(A, B)), "deterministic"
    if E == 0:
        return
    W = torch.randn(M, N, device=DEV, requires_grad=True)
    probes = [torch.randn(E, M, device=DEV), torch.randn(E, N, device=DEV), torch.randn(M, N, device=DEV)]
    outs = []
    for mm, atb in ((lambda a, w: a.matmul(w), lambda a, b: a.t().matmul(b)), (ops.MMFn.apply, ops.AtBFn.apply)):
        y = (mm(A, W) * B).sum() + (atb(A, B) / E ** 0.5).pow(2).sum()
        g1 = torch.autograd.grad(y, [A, B, W], create_graph=True)
        z = sum((x * p).sum() for x, p in zip(g1, probes))        # linear probe: well conditioned
        g2 = torch.autograd.grad(z, [A, B, W])
        outs.append(list(g1) + list(g2))
    for a, b in zip(*outs):
        close(b, a, 2e-3, 2e-4 * float(a.abs().max()) + 1e-5, "MM/AtB closure")


@pytest.mark.parametrize("G,F,E", [(30, 128, 5000), (16, 48, 333), (64, 200, 1000)])
def test_bf16_mfma_filter_variant(G, F, E):
    """bf16-operand MFMA filter (fp32 accumulate): within bf16 rounding of the fp32 network; gradients come
    from the fp32 formulas."""
    from mdgrad_amd import ops
    torch.manual_seed(G + F)
    d = (torch.rand(E, device=DEV) * 5.0).requires_grad_(True)
    mu = torch.linspace(0, 5.0, G, device=DEV)
    width = torch.full((G,), float(5.0 / (G - 1)), device=DEV)
    W1 = (torch.randn(G, G, device=DEV) / G ** 0.5).requires_grad_(True)
    b1 = torch.randn(G, device=DEV) * 0.1
    W2 = (torch.randn(F, G, device=DEV) / G ** 0.5).requires_grad_(True)
    b2 = torch.randn(F, device=DEV) * 0.1
    ref = ops.filter_reference(d, mu, width, W1, b1, W2, b2)
    got = ops.CfconvFilterFn.apply(d, mu, width, W1, b1, W2, b2, True)
    scale = float(ref.abs().max())
    close(got, ref, 0, 2e-2 * scale, "bf16 filter")            # ~2^-8 relative per operand, K <= 64 terms
    g_ref = torch.autograd.grad(ref.sum(), [d, W1, W2])
    g_got = torch.autograd.grad(got.sum(), [d, W1, W2])
    for a, b in zip(g_got, g_ref):
        close(a, b, 1e-4, 1e-5 * float(b.abs().max()), "bf16 filter gradient (fp32 formulas)")


@pytest.mark.parametrize("name", ["schnet_cg64", "schnet_water192", "schnet_cg64_wide", "schnet_cg64_a256"])
def test_analytic_schnet_passes_golden_and_autograd(name):
    """Hand-derived force / force-vjp (no autograd) against the reference goldens and the autograd path."""
    from mdgrad_amd.interface import GNNPotentials
    from mdgrad_amd.nn import get_model, analytic
    g = load_golden(name)
    system = mk_system(g["pos"], g["cell"], mass=g["masses"], numbers=g["numbers"])
    net = get_model(params_of(g))
    net.load_state_dict(sd_of(g))
    gnn = GNNPotentials(system, net, cutoff=float(g["cutoff"]))
    assert gnn.supports_force_vjp()
    q = T(g["pos"], DEV)
    gnn._reset_topology(q)
    F = gnn.force(q)
    close(F, g["F"], 1e-4, 1e-5 * np.abs(g["F"]).max(), "analytic F")
    rng = np.random.default_rng(0)
    w = T(g["w"], DEV) if "w" in g else T(rng.normal(0, 1, g["pos"].shape).astype(np.float32), DEV)
    U, F2, dq, gth = analytic.force_vjp(net, gnn._z(), q, w, gnn.inputs["_topo"], gnn.inputs["offsets"])
    close(U.reshape(1), g["U"], 1e-5, 1e-5, "U")
    close(F2, g["F"], 1e-4, 1e-5 * np.abs(g["F"]).max(), "F (vjp pass)")
    flat = torch.cat([x.reshape(-1) for x in gth])
    if "dwF_dq" in g:
        close(dq, g["dwF_dq"], 1e-3, 1e-4 * np.abs(g["dwF_dq"]).max(), "analytic d(w.F)/dq vs golden")
        close(flat, g["dwF_dtheta"], 1e-3, 1e-4 * np.abs(g["dwF_dtheta"]).max(), "analytic d(w.F)/dtheta vs golden")
    # autograd path on the same inputs
    qa = q.clone().requires_grad_(True)
    (gq,) = torch.autograd.grad(gnn(qa).sum(), qa, create_graph=True)
    plist = list(net.parameters())
    ga = torch.autograd.grad((w * -gq).sum(), [qa] + plist, allow_unused=True)
    close(dq, ga[0], 1e-3, 1e-4 * float(ga[0].abs().max()), "analytic vs autograd d(w.F)/dq")
    fa = torch.cat([(x if x is not None else torch.zeros_like(p)).reshape(-1) for x, p in zip(ga[1:], plist)])
    close(flat, fa, 1e-3, 1e-4 * float(fa.abs().max()), "analytic vs autograd d(w.F)/dtheta")


def _traj_and_grads(integ, system, t, frames_stride=1):
    from mdgrad_amd.observable import rdf
    from mdgrad_amd.sovlers import odeint_adjoint
    for p in integ.parameters():
        p.grad = None
    y0 = tuple(integ.get_inital_states(wrap=True))
    v_t, q_t, pv_t = odeint_adjoint(integ, y0, t, method="NH_verlet")
    obs = rdf(system, nbins=40, r_range=(2.0, 5.5))
    (obs(q_t[::frames_stride])[2].pow(2).mean() + 1e-3 * v_t[-1].pow(2).mean() + 1e-3 * pv_t[-1].pow(2).sum()).backward()
    gth = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in integ.parameters()])
    return v_t.detach(), q_t.detach(), pv_t.detach(), gth


@pytest.mark.gpu
@pytest.mark.parametrize("replicas", [1, 3])
def test_hip_graph_replay_equals_eager(replicas):
    """mdgrad_amd/graphs.py: forward steps and adjoint intervals replayed from captured HIP graphs (padded
    fixed-capacity neighbour lists) give the eager path's trajectory and gradients; a second pass reuses
    the graphs; too small a capacity is detected and the pass falls back to the eager path."""
    from mdgrad_amd import graphs
    g = load_golden("gnn_traj")
    base = mk_system(g["pos"], g["cell"], g["vel"], g["masses"], g["numbers"])
    system = base
    if replicas > 1:
        rng = np.random.default_rng(2)
        system = base.replicate(replicas)
        system.set_positions(np.concatenate([np.mod(g["pos"] + rng.normal(0, 0.05, g["pos"].shape), g["cell"])
                                             for _ in range(replicas)]))
        system.set_velocities(np.concatenate([g["vel"] * (1 + 0.1 * r) for r in range(replicas)]))
    t = torch.Tensor([float(g["dt"]) * i for i in range(7)]).to(DEV)
    integ = _gnn_integrator(g, system)
    assert graphs.enabled(integ)
    integ.use_graphs = False
    ref = _traj_and_grads(integ, system, t)
    integ.use_graphs = True
    out = _traj_and_grads(integ, system, t)
    assert len(integ._graph_cache) == 2, "forward and adjoint graphs captured"
    for a, b, name in zip(out, ref, ("v_t", "q_t", "pv_t", "dL/dtheta")):
        close(a, b, 1e-4, 2e-5 * float(b.abs().max()) + 1e-7, "graph vs eager " + name)
    cached = dict(integ._graph_cache)
    out2 = _traj_and_grads(integ, system, t)
    assert all(integ._graph_cache[k] is v for k, v in cached.items()), "graphs reused"
    for a, b in zip(out2, out):
        assert torch.equal(a, b), "replay is reproducible"
    # a new thermostat temperature (annealing schedules call update_T every epoch) reuses the graphs
    T0 = integ.T
    integ.update_T(1.7 * T0)
    hot = _traj_and_grads(integ, system, t)
    assert all(integ._graph_cache[k] is v for k, v in cached.items()), "graphs survive update_T"
    integ.use_graphs = False
    hot_ref = _traj_and_grads(integ, system, t)
    integ.use_graphs = True
    for a, b, name in zip(hot, hot_ref, ("v_t", "q_t", "pv_t", "dL/dtheta")):
        close(a, b, 1e-4, 2e-5 * float(b.abs().max()) + 1e-7, "graph vs eager after update_T " + name)
    assert float((hot[2] - out[2]).abs().max()) > 0, "the new temperature reached the captured kernels"
    integ.update_T(T0)
    # exact-size lists again outside the graphed passes (the autograd path must not see padding rows)
    gnn = integ.model.models["gnn"]
    assert not gnn._static_on and int(gnn.inputs["nbr_list"].min()) >= 0
    # capacity overflow: shrink the edge capacity below the current pair count
    gnn._static["capacity"] = 256
    gnn._static["version"] += 1
    out3 = _traj_and_grads(integ, system, t)
    # (the forward pass overflowed, fell back to the eager loop and dropped its graph; the adjoint pass
    #  that followed captured a new graph with the enlarged capacity)
    assert gnn._static["capacity"] > 256 and [k[0] for k in integ._graph_cache] == ["adj"]
    for a, b, name in zip(out3, ref, ("v_t", "q_t", "pv_t", "dL/dtheta")):
        close(a, b, 1e-4, 2e-5 * float(b.abs().max()) + 1e-7, "overflow fallback " + name)
    out4 = _traj_and_grads(integ, system, t)                    # re-captured with the enlarged capacity
    assert len(integ._graph_cache) == 2
    for a, b, name in zip(out4, ref, ("v_t", "q_t", "pv_t", "dL/dtheta")):
        close(a, b, 1e-4, 2e-5 * float(b.abs().max()) + 1e-7, "after regrow " + name)


@pytest.mark.gpu
def test_nve_generic_analytic_adjoint_equals_autograd():
    """NVE (torchmd/md.py:98-157) over GNN + prior on the generic path: the analytic-adjoint protocol
    (NVE.rhs_vjp, the verlet backward branch of sovlers.py:42-101 unchanged) against the reference's
    autograd double backward."""
    from mdgrad_amd import potentials as P
    from mdgrad_amd.interface import GNNPotentials, PairPotentials, Stack
    from mdgrad_amd.md import NVE
    from mdgrad_amd.nn import get_model
    from mdgrad_amd.sovlers import odeint_adjoint
    g = load_golden("gnn_traj")
    system = mk_system(g["pos"], g["cell"], g["vel"], g["masses"], g["numbers"])
    net = get_model(params_of(g))
    net.load_state_dict(sd_of(g))
    gnn = GNNPotentials(system, net, cutoff=float(g["cutoff"]))
    prior = PairPotentials(system, P.ExcludedVolume(float(g["prior_sigma"]), float(g["prior_epsilon"]), 12),
                           cutoff=float(g["cutoff"]))
    integ = NVE(Stack({"gnn": gnn, "prior": prior}), system).to(DEV)
    t = torch.Tensor([float(g["dt"]) * i for i in range(6)]).to(DEV)
    params = list(integ.parameters())

    def run(analytic):
        gnn.analytic = analytic
        assert integ.supports_rhs_vjp() == analytic
        for p_ in params:
            p_.grad = None
        y0 = [s_.clone().requires_grad_(True) for s_ in integ.get_inital_states(wrap=True)]
        v_t, q_t = odeint_adjoint(integ, tuple(y0), t, method="verlet")
        (q_t[-1].pow(2).mean() + v_t[::2].pow(2).mean()).backward()
        gth = torch.cat([(p_.grad if p_.grad is not None else torch.zeros_like(p_)).reshape(-1) for p_ in params])
        return q_t.detach(), y0[0].grad, y0[1].grad, gth

    a, b = run(True), run(False)
    close(a[0], b[0], 1e-4, 1e-5, "q_t")
    close(a[1], b[1], 2e-3, 1e-4 * float(b[1].abs().max()), "dL/dv0")
    close(a[2], b[2], 2e-3, 1e-4 * float(b[2].abs().max()), "dL/dq0")
    close(a[3], b[3], 5e-3, 2e-4 * float(b[3].abs().max()), "dL/dtheta")


@pytest.mark.gpu
@pytest.mark.parametrize("frames", [3, 7])
def test_nve_analytic_verlet_paths_equal_the_generic_solver(frames):
    """NVE over GNN + prior: the cached-force forward integration and the analytic adjoint of `verlet` (Verlet.integrate,
    sovlers._analytic_nve_adjoint: two force-vjp evaluations per interval) -- eagerly (3 frames) and replayed from HIP graphs
    (7 frames; once more with graphs switched off) -- against the reference's own control flow on the same kernels: the
    two-call step and the 6-state backward branch of verlet_update through the generic solver (sovlers.py:21-104, :196-293)."""
    from mdgrad_amd import potentials as P
    from mdgrad_amd.interface import GNNPotentials, PairPotentials, Stack
    from mdgrad_amd.md import NVE
    from mdgrad_amd.nn import get_model
    from mdgrad_amd.sovlers import odeint_adjoint
    g = load_golden("gnn_traj")
    system = mk_system(g["pos"], g["cell"], g["vel"], g["masses"], g["numbers"])
    net = get_model(params_of(g))
    net.load_state_dict(sd_of(g))
    gnn = GNNPotentials(system, net, cutoff=float(g["cutoff"]))
    prior = PairPotentials(system, P.ExcludedVolume(float(g["prior_sigma"]), float(g["prior_epsilon"]), 12),
                           cutoff=float(g["cutoff"]))
    integ = NVE(Stack({"gnn": gnn, "prior": prior}), system).to(DEV)
    t = torch.Tensor([float(g["dt"]) * i for i in range(frames)]).to(DEV)
    params = list(integ.parameters())

    def run(analytic, graphs_on=True):
        integ.analytic_verlet, integ.use_graphs = analytic, graphs_on
        for p_ in params:
            p_.grad = None
        y0 = [s_.clone().requires_grad_(True) for s_ in integ.get_inital_states(wrap=True)]
        v_t, q_t = odeint_adjoint(integ, tuple(y0), t, method="verlet")
        (q_t[-1].pow(2).mean() + v_t[::2].pow(2).mean() + q_t[1].sum() * 1e-3).backward()
        gth = torch.cat([(p_.grad if p_.grad is not None else torch.zeros_like(p_)).reshape(-1) for p_ in params])
        return v_t.detach(), q_t.detach(), y0[0].grad, y0[1].grad, gth

    ref = run(False)
    assert not getattr(integ, "_graph_cache", None), "the generic solver captures nothing"
    outs = [run(True)]
    if frames > 3:
        assert len(integ._graph_cache) == 2, "forward and adjoint graphs of the verlet steps"
        outs.append(run(True))                                     # replay of the cached graphs
        outs.append(run(True, graphs_on=False))
    for o in outs:
        for a, b, name in zip(o, ref, ("v_t", "q_t", "dL/dv0", "dL/dq0", "dL/dtheta")):
            close(a, b, 1e-5, 1e-6 * float(b.abs().max()) + 1e-9, "analytic verlet vs generic solver: " + name)


@pytest.mark.gpu
def test_fit_rdf_gnn_example_runs():
    """examples/fit_rdf_gnn.py (the loop of demo/fit_rdf_gnn.py:380-461: annealing through update_T,
    Simulations epochs, JS / compute_D losses, Adam + ReduceLROnPlateau) on a small stacked system."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "fit_rdf_gnn", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples",
                                    "fit_rdf_gnn.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    hist = mod.main(["--size", "3", "--replicas", "4", "--epochs", "5", "--tau", "40"])
    assert len(hist) == 5 and all(np.isfinite(h[0]) and np.isfinite(h[1]) for h in hist)
    assert hist[0][2] > hist[-1][2] >= 298.0, "annealing schedule applied"
    assert hist[-1][0] < hist[0][0]


@pytest.mark.gpu
def test_trainable_gaussian_basis_trajectory_graph_replay_equals_eager_and_moves_the_basis():
    """`trainable_gauss=True` through the whole path: Stack(SchNet + prior), NHC, RDF loss, analytic adjoint -- replayed from
    captured HIP graphs and eagerly on exact lists -- same trajectory, same gradients, non-zero gradients on `width` and
    `offsets`; after an optimizer step the replayed graphs see the new basis (they read the live parameters)."""
    from mdgrad_amd import graphs, potentials as P
    from mdgrad_amd.interface import GNNPotentials, PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    from mdgrad_amd.nn import get_model
    g = load_golden("gnn_traj")
    system = mk_system(g["pos"], g["cell"], g["vel"], g["masses"], g["numbers"])
    torch.manual_seed(3)
    net = get_model(dict(params_of(g), trainable_gauss=True))
    gnn = GNNPotentials(system, net, cutoff=float(g["cutoff"]))
    prior = PairPotentials(system, P.ExcludedVolume(float(g["prior_sigma"]), float(g["prior_epsilon"]), 12), cutoff=float(g["cutoff"]))
    integ = NoseHooverChain(Stack({"gnn": gnn, "prior": prior}), system, T=float(g["T"]), num_chains=int(g["chains"]),
                            Q=float(g["Q"]), adjoint=True).to(DEV)
    assert gnn.supports_force_vjp() and graphs.enabled(integ)
    t = torch.Tensor([float(g["dt"]) * i for i in range(7)]).to(DEV)
    names = [n for n, _ in integ.named_parameters()]
    basis = [k for k, n in enumerate(names) if n.endswith(".0.width") or n.endswith(".0.offsets")]
    assert len(basis) == 2 * len(net.convolutions)
    out = _traj_and_grads(integ, system, t)
    integ.use_graphs = False
    ref = _traj_and_grads(integ, system, t)
    integ.use_graphs = True
    for a, b, name in zip(out, ref, ("v_t", "q_t", "pv_t", "dL/dtheta")):
        close(a, b, 1e-4, 2e-5 * float(b.abs().max()) + 1e-7, "trainable basis, graph vs eager " + name)
    sizes = np.cumsum([0] + [p.numel() for p in integ.parameters()])
    for k in basis:
        assert float(out[3][sizes[k]:sizes[k + 1]].abs().max()) > 0, "no gradient on " + names[k]
    with torch.no_grad():
        for conv in net.convolutions:
            conv.moduledict["message_edge_filter"][0].offsets.add_(0.05)
    moved = _traj_and_grads(integ, system, t)
    integ.use_graphs = False
    moved_ref = _traj_and_grads(integ, system, t)
    close(moved[1], moved_ref[1], 1e-4, 2e-5 * float(moved_ref[1].abs().max()), "after moving the centres: q_t")
    close(moved[3], moved_ref[3], 1e-4, 2e-5 * float(moved_ref[3].abs().max()) + 1e-7, "after moving the centres: dL/dtheta")
    assert float((moved[1] - out[1]).abs().max()) > 0
    # ... and the new WIDTHS: the kernels read -0.5 / width^2 from a derived buffer (analytic._gauss_coeff), which a replayed
    # graph must see refreshed (it used to bake in a cached tensor of the capture's warm-up runs)
    integ.use_graphs = True
    with torch.no_grad():
        for conv in net.convolutions:
            conv.moduledict["message_edge_filter"][0].width.mul_(1.3)
    wide = _traj_and_grads(integ, system, t)
    integ.use_graphs = False
    wide_ref = _traj_and_grads(integ, system, t)
    close(wide[1], wide_ref[1], 1e-4, 2e-5 * float(wide_ref[1].abs().max()), "after widening the Gaussians: q_t")
    close(wide[3], wide_ref[3], 1e-4, 2e-5 * float(wide_ref[3].abs().max()) + 1e-7, "after widening the Gaussians: dL/dtheta")
    assert float((wide[1] - moved[1]).abs().max()) > 1e-6 * float(moved[1].abs().max()), "the replayed steps ignored the new widths"
