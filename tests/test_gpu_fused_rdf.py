"""The RDF observable evaluated INSIDE the wave-per-replica trajectory kernels (csrc/traj_ring.hpp, RDF = 1 / 2;
mdg_traj_fwd_small_rdf / mdg_traj_adj_small_rdf) against the separate observable launches it replaces
(ops.RdfRawFn on the stored frames, themselves pinned to the oracle in tests/test_gpu_pins.py) and against the CPU
oracle directly: histogram, g(r), and every gradient that flows through it -- d/d(sigma, epsilon), d/d(v0, q0, pv0).

The fusion is opt-in (`integrator.fuse_observables = True` or `attach_observable`): `rdf.forward` called on a fused
trajectory then registers itself with the integrator, the NEXT launch produces the histogram; a slice along time that runs to the last frame (q_t[::k], q_t[s:]) is recognised."""
import numpy as np
import pytest
import torch

import oracle as O
from conftest import load_golden
from test_gpu_parity import T, close, mk_system, lj_setup, oracle_run, DEV

pytestmark = pytest.mark.gpu


def _setup(n_atoms=108, ensemble="nhc", R=3, seed=0, nT=9):
    from mdgrad_amd import potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NVE, NoseHooverChain
    from mdgrad_amd.observable import rdf
    g = load_golden("nhc_traj_lj")
    base = g["pos"][:n_atoms]
    system = mk_system(base, g["cell"], g["vel"][:n_atoms], g["mass"][:n_atoms])
    mdl = P.LennardJones(1.0, 1.0)
    stack = Stack({"pair": PairPotentials(system, mdl, cutoff=2.5)})
    nhc = ensemble == "nhc"
    integ = (NoseHooverChain(stack, system, T=1.0, num_chains=5, Q=50.0) if nhc else NVE(stack, system)).to(DEV)
    integ.fuse_observables = True                        # (opt-in since round 5; the default is tested at the end of this file)
    spec = integ.fused_spec("NH_verlet" if nhc else "verlet")
    spec.block = 64
    rng = np.random.default_rng(seed)
    pos = np.mod(base[None] + rng.normal(0, 0.03, (R,) + base.shape), g["cell"]).astype(np.float32)
    vel = rng.normal(0, 1.0, pos.shape).astype(np.float32)
    t = torch.Tensor([0.005 * i for i in range(nT)]).to(DEV)
    obs = rdf(system, nbins=100, r_range=(0.75, 2.5))
    return g, mdl, integ, spec, pos, vel, t, obs, nhc


def _run(mdl, spec, pos, vel, t, obs, nhc, pick, extra=False):
    """One forward + backward; returns g, the gradients and whether the launch was fused."""
    from mdgrad_amd import ops
    R = pos.shape[0]
    v0, q0 = T(vel, DEV).requires_grad_(True), T(pos, DEV).requires_grad_(True)
    pv0 = torch.zeros(R, 5, device=DEV, requires_grad=True) if nhc else None
    out = ops.fused_traj(v0, q0, pv0, t, spec.flat_params(), spec)
    q_t = out[1]
    fused = q_t._mdg_traj[3] is not None
    _, _, gr = obs(pick(q_t))
    wgt = torch.linspace(0.5, 1.5, gr.shape[0], device=DEV)
    loss = (gr * wgt).pow(2).sum()
    if extra:                                            # other consumers of the trajectory beside the observable
        loss = loss + q_t[:, ::2].pow(2).sum() / 100.0 + out[0][:, -1].pow(2).sum() / 50.0
        if nhc:
            loss = loss + out[2][:, -1].sum()
    mdl.zero_grad()
    loss.backward()
    grads = [v0.grad, q0.grad] + ([pv0.grad] if nhc else []) + [mdl.sigma.grad.clone(), mdl.epsilon.grad.clone()]
    return gr.detach(), [x.detach().clone() for x in grads], fused, q_t.detach()


@pytest.mark.parametrize("n_atoms,ensemble,pick_name,extra", [
    (108, "nhc", "all", False), (108, "nhc", "all", True), (108, "nhc", "stride3", True), (108, "nhc", "from2", False),
    (108, "nve", "all", True), (107, "nhc", "stride2from1", False), (31, "nve", "all", False)])
def test_fused_rdf_equals_separate_observable_launches(n_atoms, ensemble, pick_name, extra):
    picks = {"all": lambda q: q, "stride3": lambda q: q[:, ::3], "from2": lambda q: q[:, 2:],
             "stride2from1": lambda q: q[:, 1::2]}
    g, mdl, integ, spec, pos, vel, t, obs, nhc = _setup(n_atoms, ensemble)
    pick = picks[pick_name]
    ref = _run(mdl, spec, pos, vel, t, obs, nhc, pick, extra)           # registers the observable
    assert not ref[2], "the first launch knows nothing about the observable"
    fus = _run(mdl, spec, pos, vel, t, obs, nhc, pick, extra)
    assert fus[2], "the second launch must produce the histogram itself"
    assert torch.equal(ref[3], fus[3]), "the trajectory itself does not change"
    # (27 frames: the separate launch runs the exact lane-private kernels, the fused one always counts on the fine
    #  integer grid, whose binning error only averages down to the 2e-5 of tests/test_gpu_pins.py over the >= 1024
    #  frames it is meant for -- test_fused_rdf_histogram_is_bitwise_the_many_frame_kernels covers that regime.
    #  dL/dg is proportional to g here, so the gradients inherit the same relative error.)
    close(fus[0], ref[0], 2e-4, 1e-4, "g(r)")
    names = ["adj v0", "adj q0"] + (["adj pv0"] if nhc else []) + ["dsigma", "depsilon"]
    for a, b, nm in zip(fus[1], ref[1], names):
        close(a, b, 1e-3, 2e-4 * float(b.abs().max()) + 1e-7, nm)
    # a selection that does not run to the last frame is not fusable: computed the plain way, still right
    cut = _run(mdl, spec, pos, vel, t, obs, nhc, lambda q: q[:, :5], extra)
    from mdgrad_amd import ops
    with torch.no_grad():
        plain = obs(torch.as_tensor(fus[3][:, :5]).clone())[2]
    close(cut[0], plain, 1e-6, 1e-6, "unfusable slice")


def test_fused_rdf_vs_oracle():
    """Fused launch against the oracle end to end: g(r) of all frames of 3 replicas, d(loss)/d(sigma, epsilon) and the
    adjoints of the initial state."""
    g, mdl, integ, spec, pos, vel, t, obs, nhc = _setup(108, "nhc", R=3, nT=9)
    _run(mdl, spec, pos, vel, t, obs, nhc, lambda q: q)
    gr, grads, fused, q_t = _run(mdl, spec, pos, vel, t, obs, nhc, lambda q: q)
    assert fused
    wgt = torch.linspace(0.5, 1.5, 100)
    term = O.PairTerm("lj", torch.tensor([1.0, 1.0]), 2.5, T(g["cell"]), p=12, q=6, c=1)
    model = O.ModelOracle([term])
    eom = O.NHCOracle(model, T(g["mass"]), 1.0, 50.0, 5)
    trajs = [O.odeint_oracle(eom, (T(vel[r]), T(pos[r]), torch.zeros(5)), t.cpu()) for r in range(3)]
    leaves = [[x.detach().clone().requires_grad_(True) for x in tr] for tr in trajs]
    frames = torch.cat([lv[1] for lv in leaves])
    raw = O.rdf_raw_oracle(frames, T(g["cell"]), 100, (0.75, 2.5))
    _, _, go = O.rdf_normalise_oracle(raw, 100, (0.75, 2.5))
    (go * wgt).pow(2).sum().backward()
    close(gr, go.detach(), 2e-4, 1e-4, "g(r) vs oracle")          # (27 frames on the fine grid, see above)
    gth = np.zeros(2)
    for r in range(3):
        lam, gt = O.adjoint_oracle(eom, trajs[r], [x.grad if x.grad is not None else torch.zeros_like(x) for x in leaves[r]], t.cpu())
        close(grads[0][r], lam[0], 1e-3, 2e-4 * float(lam[0].abs().max()), "adj v0[%d]" % r)
        close(grads[1][r], lam[1], 1e-3, 2e-4 * float(lam[1].abs().max()), "adj q0[%d]" % r)
        gth += gt.numpy()
    got = np.array([float(grads[3]), float(grads[4])])
    close(got, gth, 1e-3, 2e-4 * np.abs(gth).max(), "dL/dtheta vs oracle")


def test_fused_rdf_histogram_is_bitwise_the_many_frame_kernels():
    """>= 1024 frames: the separate forward is the fine integer histogram too (rdf_fwd_fine_kernel) -- same grid,
    same integer counts, so the raw histogram must agree to the last bit, and two fused launches with each other."""
    from mdgrad_amd import ops
    g, mdl, integ, spec, pos, vel, t, obs, nhc = _setup(108, "nhc", R=128, nT=9)
    v0, q0 = T(vel, DEV), T(pos, DEV)
    pv0 = torch.zeros(128, 5, device=DEV)
    with torch.no_grad():
        q_a = ops.fused_traj(v0, q0, pv0, t, spec.flat_params(), spec)[1]
        raw_plain = ops.RdfRawFn.apply(q_a, obs.offsets, obs.coeff, obs.cutoff_boundary, obs._cell_struct, None, obs.spacing)
        obs(q_a)                                                       # registers
        outs = [ops.fused_traj(v0, q0, pv0, t, spec.flat_params(), spec)[1]._mdg_traj[3] for _ in range(2)]
    assert outs[0] is not None and torch.equal(outs[0], outs[1])
    assert torch.equal(outs[0], raw_plain)


def test_fused_rdf_time_gradients_and_opt_out():
    """A time grid that requires grad makes the backward materialise the observable's frame gradients (time_vjps,
    sovlers.py:258-266, read them): same numbers as the unfused run.  `fuse_observables = False` on the integrator
    keeps the separate launches."""
    from mdgrad_amd import ops
    g, mdl, integ, spec, pos, vel, t0, obs, nhc = _setup(108, "nhc", R=2, nT=7)
    res = []
    for rep in range(2):
        t = t0.clone().requires_grad_(True)
        v0, q0 = T(vel, DEV).requires_grad_(True), T(pos, DEV).requires_grad_(True)
        pv0 = torch.zeros(2, 5, device=DEV, requires_grad=True)
        v_t, q_t, pv_t = ops.fused_traj(v0, q0, pv0, t, spec.flat_params(), spec)
        _, _, gr = obs(q_t[:, 1::2])
        mdl.zero_grad()
        (gr.pow(2).sum() + v_t[:, -1].pow(2).sum()).backward()
        res.append([gr.detach(), t.grad.clone(), q0.grad.clone(), mdl.sigma.grad.clone(), q_t._mdg_traj[3] is not None])
    assert not res[0][4] and res[1][4]
    for a, b, nm in zip(res[1][:4], res[0][:4], ["g", "dL/dt", "adj q0", "dsigma"]):
        close(a, b, 1e-3, 2e-4 * float(b.abs().max()) + 1e-7, nm)
    # opt-out: specs made from now on carry no hint
    integ.fuse_observables = False
    assert integ.fused_spec("NH_verlet").rdf_hint is None
    integ.fuse_observables = True
    assert integ.fused_spec("NH_verlet").rdf_hint is not None


def test_single_replica_trajectories_keep_the_plain_observable():
    """odeint_adjoint on one replica runs the LDS-resident kernels (latency, not throughput): the observable
    registers, nothing is fused, results repeat."""
    from mdgrad_amd.sovlers import odeint_adjoint
    from mdgrad_amd.observable import rdf
    g = load_golden("nhc_traj_lj")
    system, mdl, integ = lj_setup(g)
    obs = rdf(system, nbins=100, r_range=(0.75, 2.5))
    res = []
    for rep in range(2):
        y0 = [s.clone().requires_grad_(True) for s in integ.get_inital_states(wrap=True)]
        t = torch.Tensor([0.005 * i for i in range(12)]).to(DEV)
        v_t, q_t, pv_t = odeint_adjoint(integ, tuple(y0), t, method="NH_verlet")
        _, _, gr = obs(q_t[::2])
        mdl.zero_grad()
        (gr.pow(2).sum() + v_t[-1].pow(2).sum()).backward()
        res.append([gr.detach(), y0[1].grad.clone(), mdl.sigma.grad.clone()])
        assert q_t._mdg_traj[3] is None
    for a, b in zip(res[1], res[0]):
        assert torch.equal(a, b)


def test_fused_rdf_through_the_reference_api_with_stacked_replicas():
    """System.replicate(R) + Simulations.simulate + rdf(q_t[::k]) -- the reference's own call sequence on a
    replica-stacked system ([T, R*N, 3] trajectories): the second pass is fused, its g(r) and parameter gradients
    equal the unfused ones."""
    from mdgrad_amd import potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain, Simulations
    from mdgrad_amd.observable import rdf
    g = load_golden("nhc_traj_lj")
    R = 1024
    base = mk_system(g["pos"], g["cell"], g["vel"], g["mass"])
    system = base.replicate(R)
    rng = np.random.default_rng(11)
    pos0 = np.mod(np.tile(g["pos"], (R, 1)) + rng.normal(0, 0.03, (R * 108, 3)), g["cell"]).astype(np.float32)
    vel0 = rng.normal(0, 1.0, pos0.shape).astype(np.float32)
    mdl = P.LennardJones(1.0, 1.0)
    integ = NoseHooverChain(Stack({"pair": PairPotentials(system, mdl, cutoff=2.5)}), system, T=1.0, num_chains=5,
                            Q=50.0).to(DEV)
    obs = rdf(system, nbins=100, r_range=(0.75, 2.5))
    res = {}
    for fuse in (False, True, True):
        integ.fuse_observables = fuse
        system.set_positions(pos0)
        system.set_velocities(vel0)
        sim = Simulations(system, integ)
        v_t, q_t, pv_t = sim.simulate(steps=10, frequency=10, dt=0.005)
        assert q_t.shape == (10, R * 108, 3)
        _, _, gr = obs(q_t[::3])
        mdl.zero_grad()
        (gr - 1).pow(2).mean().backward()
        res[fuse] = (gr.detach().clone(), mdl.sigma.grad.clone(), mdl.epsilon.grad.clone(), q_t._mdg_traj[3] is not None)
    assert res[True][3], "the pass after the registering one must be fused"
    ref = None
    integ.fuse_observables = False
    system.set_positions(pos0)
    system.set_velocities(vel0)
    q_ref = Simulations(system, integ).simulate(steps=10, frequency=10, dt=0.005)[1]
    _, _, gr = obs(q_ref[::3])
    mdl.zero_grad()
    (gr - 1).pow(2).mean().backward()
    assert q_ref._mdg_traj[3] is None
    close(res[True][0], gr, 2e-5, 2e-5, "g(r): fused vs separate, 4096 frames")
    close(res[True][1], mdl.sigma.grad, 1e-3, 1e-6, "dsigma")
    close(res[True][2], mdl.epsilon.grad, 1e-3, 1e-6, "depsilon")


def test_fused_rdf_reports_its_path_and_does_not_pass_through_q_t():
    """the automatic fusion changes the autograd graph -- from the second pass on the histogram is an output
    of the trajectory launch, not a function of q_t.  The observable says which path ran (`last_path`), warns once
    when it switches, and a derivative w.r.t. q_t that would silently lose the RDF term raises instead."""
    import warnings
    from mdgrad_amd import ops
    g, mdl, integ, spec, pos, vel, t, obs, nhc = _setup(108, "nhc", R=2, nT=7)
    paths = []
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        for rep in range(3):
            v0, q0 = T(vel, DEV), T(pos, DEV)
            pv0 = torch.zeros(2, 5, device=DEV)
            v_t, q_t, pv_t = ops.fused_traj(v0, q0, pv0, t, spec.flat_params(), spec)
            _, _, gr = obs(q_t)
            paths.append(obs.last_path)
    assert paths == ["kernel", "fused-trajectory", "fused-trajectory"]
    assert sum("fused trajectory launch" in str(w.message) for w in rec) == 1          # once, not per pass
    with pytest.raises(RuntimeError):                    # the RDF term does not reach q_t in the graph: no silent zero
        torch.autograd.grad(gr.pow(2).sum(), q_t)


def test_fusion_is_off_by_default_and_q_t_carries_the_rdf_term():
    """without the opt-in nothing is fused, however often the observable is called on fused
    trajectories -- the histogram stays a function of q_t in the autograd graph, so d(loss)/d(q_t) is the RDF term (as
    for a reference caller) and equals the oracle's."""
    from mdgrad_amd import ops, potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    from mdgrad_amd.observable import rdf
    g = load_golden("nhc_traj_lj")
    system = mk_system(g["pos"], g["cell"], g["vel"], g["mass"])
    mdl = P.LennardJones(1.0, 1.0)
    integ = NoseHooverChain(Stack({"pair": PairPotentials(system, mdl, cutoff=2.5)}), system, T=1.0, num_chains=5, Q=50.0).to(DEV)
    assert integ.fuse_observables is False
    spec = integ.fused_spec("NH_verlet")
    spec.block = 64
    obs = rdf(system, nbins=100, r_range=(0.75, 2.5))
    rng = np.random.default_rng(3)
    pos = np.mod(g["pos"][None] + rng.normal(0, 0.03, (2,) + g["pos"].shape), g["cell"]).astype(np.float32)
    vel = rng.normal(0, 1.0, pos.shape).astype(np.float32)
    t = torch.Tensor([0.005 * i for i in range(6)]).to(DEV)
    for rep in range(3):
        v_t, q_t, pv_t = ops.fused_traj(T(vel, DEV), T(pos, DEV), torch.zeros(2, 5, device=DEV), t, spec.flat_params(), spec)
        _, _, gr = obs(q_t)
        assert obs.last_path == "kernel" and q_t._mdg_traj[3] is None and spec.rdf_hint is None
    (gq,) = torch.autograd.grad(gr.pow(2).sum(), q_t)
    xo = q_t.detach().cpu().reshape(-1, 108, 3).clone().requires_grad_(True)
    _, _, go = O.rdf_oracle(xo, T(g["cell"]), 100, (0.75, 2.5))
    (gxo,) = torch.autograd.grad(go.pow(2).sum(), xo)
    close(gq.reshape(-1, 108, 3), gxo, 1e-3, 1e-4 * float(gxo.abs().max()), "d(loss)/d(q_t) through the observable")
    integ.attach_observable(obs)                          # the explicit opt-in
    assert integ.fuse_observables is True and integ.fused_spec("NH_verlet").rdf_hint is not None
