"""The kernel variants that bench.py actually times, pinned to the CPU oracle (which is itself pinned to the
reference goldens, tests/test_oracle_golden.py):

  * the one-lane-per-atom trajectory kernels (workgroup = 128 threads; picked when >= 1024 replicas run in one
    launch, csrc/traj_small.hip pick_block) -- forward, adjoint and parameter gradients;
  * the many-frame (>= 1024 frames) RDF kernels: half-width and full-width lane-per-pair forward, fine-table and
    recurrence backward (csrc/rdf.hip);
  * the multi-launch large-N trajectory kernels (csrc/traj_large.hip) at 1 000, 2 744 and 4 096 atoms;
  * BASELINE config #3 (192-atom water, SchNet + prior) as a trajectory + adjoint golden from the reference.

fp32 tolerances are written at each assert (about 10x the errors observed on MI355X)."""
import numpy as np
import pytest
import torch

import oracle as O
from conftest import load_golden
from test_gpu_parity import T, close, mk_system, lj_setup, oracle_run, liquid, DEV

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ timed trajectory geometry
@pytest.mark.parametrize("R,block", [(6, 128), (6, 64), (1024, 0)])
def test_one_lane_per_atom_trajectory_kernels_vs_oracle(R, block):
    """block = 128 (one lane per atom, LDS-resident) and block = 64 (wave per replica, register-resident ring
    sweep, csrc/traj_ring.hpp) forced on a few replicas, and the default launch at R = 1024 (which picks the ring
    kernels for LJ 12-6): sampled replicas' trajectories, adjoints w.r.t. the initial state and the summed
    parameter gradient == oracle."""
    from mdgrad_amd import ops
    g = load_golden("nhc_traj_lj")
    system, mdl, integ = lj_setup(g)
    spec = integ.fused_spec("NH_verlet")
    spec.block = block
    nT = 12
    rng = np.random.default_rng(7 + R)
    pos = np.mod(g["pos"][None] + rng.normal(0, 0.03, (R,) + g["pos"].shape), g["cell"]).astype(np.float32)
    vel = rng.normal(0, 1.0, pos.shape).astype(np.float32)
    t = torch.Tensor([0.005 * i for i in range(nT)])
    sample = [0, 1, R // 2, R - 1]
    wr = torch.zeros(R, device=DEV)
    wr[sample] = 1.0                                     # only the sampled replicas feed the loss
    v0, q0 = T(vel, DEV).requires_grad_(True), T(pos, DEV).requires_grad_(True)
    pv0 = torch.zeros(R, 5, device=DEV, requires_grad=True)
    v_t, q_t, pv_t = ops.FusedTrajFn.apply(v0, q0, pv0, t.to(DEV), spec.flat_params(), spec)
    mdl.zero_grad()
    loss = ((q_t[:, ::2].pow(2).sum((1, 2, 3)) / (6 * 108 * 3) + v_t[:, -1].pow(2).sum((1, 2)) / (108 * 3)
             + pv_t[:, -1].sum(1)) * wr).sum()
    loss.backward()
    gth_sum = np.zeros(2)
    for r in sample:
        term = O.PairTerm("lj", torch.tensor([1.0, 1.0]), 2.5, T(g["cell"]), p=12, q=6, c=1)
        traj, lam, gth = oracle_run(
            pos[r], g["cell"], vel[r], g["mass"], [term], 1.0, 50.0, 5, t,
            lambda L: L[1][::2].pow(2).sum() / (6 * 108 * 3) + L[0][-1].pow(2).sum() / (108 * 3) + L[2][-1].sum())
        close(q_t[r], traj[1], 0, 2e-5, "q_t[%d]" % r)
        close(v_t[r], traj[0], 0, 2e-4, "v_t[%d]" % r)
        close(pv_t[r], traj[2], 1e-4, 1e-4, "pv_t[%d]" % r)
        close(v0.grad[r], lam[0], 1e-3, 2e-4 * float(lam[0].abs().max()), "adj v0[%d]" % r)
        close(q0.grad[r], lam[1], 1e-3, 2e-4 * float(lam[1].abs().max()), "adj q0[%d]" % r)
        close(pv0.grad[r], lam[2], 1e-3, 2e-4 * float(lam[2].abs().max()) + 1e-6, "adj pv0[%d]" % r)
        gth_sum += gth.numpy()
    got = np.array([float(mdl.sigma.grad), float(mdl.epsilon.grad)])
    close(got, gth_sum, 1e-3, 2e-4 * np.abs(gth_sum).max(), "sum over sampled replicas of dL/dtheta")
    rest = [r for r in range(R) if r not in sample]
    assert float(q0.grad[rest].abs().max()) == 0.0, "replicas outside the loss get exactly zero adjoint"


@pytest.mark.parametrize("n_atoms,ensemble", [(108, "nve"), (107, "nhc"), (31, "nve"), (54, "nhc"), (3, "nhc"),
                                              (2, "nve"), (108, "nhc-exvol"), (53, "nve-exvol")])
def test_ring_kernels_odd_sizes_and_nve_vs_oracle(n_atoms, ensemble):
    """The wave-per-replica ring kernels (block = 64) for ring lengths that are odd / even / tiny, a lane with one
    atom, and the NVE branch of the adjoint (sovlers.py:42-101): trajectory, adjoints and parameter gradient of 3
    replicas == oracle."""
    from mdgrad_amd import ops, potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NVE, NoseHooverChain
    g = load_golden("nhc_traj_lj")
    R, nT = 3, 9
    exvol = ensemble.endswith("-exvol")               # ExcludedVolume(power 12): the same kernels with c = 0
    ensemble = ensemble.split("-")[0]
    rng = np.random.default_rng(n_atoms)
    base = g["pos"][:n_atoms]
    system = mk_system(base, g["cell"], g["vel"][:n_atoms], g["mass"][:n_atoms])
    mdl = P.ExcludedVolume(1.0, 1.0, 12) if exvol else P.LennardJones(1.0, 1.0)
    stack = Stack({"pair": PairPotentials(system, mdl, cutoff=2.5)})
    nhc = ensemble == "nhc"
    integ = (NoseHooverChain(stack, system, T=1.0, num_chains=5, Q=50.0) if nhc else NVE(stack, system)).to(DEV)
    spec = integ.fused_spec("NH_verlet" if nhc else "verlet")
    spec.block = 64
    pos = np.mod(base[None] + rng.normal(0, 0.03, (R,) + base.shape), g["cell"]).astype(np.float32)
    vel = rng.normal(0, 1.0, pos.shape).astype(np.float32)
    t = torch.Tensor([0.005 * i for i in range(nT)])
    v0, q0 = T(vel, DEV).requires_grad_(True), T(pos, DEV).requires_grad_(True)
    pv0 = torch.zeros(R, 5, device=DEV, requires_grad=True) if nhc else None
    out = ops.FusedTrajFn.apply(v0, q0, pv0, t.to(DEV), spec.flat_params(), spec)
    v_t, q_t = out[0], out[1]
    mdl.zero_grad()
    nrm = n_atoms * 3
    loss = (q_t[:, ::2].pow(2).sum((1, 2, 3)) / (5 * nrm) + v_t[:, -1].pow(2).sum((1, 2)) / nrm).sum()
    if nhc:
        loss = loss + out[2][:, -1].sum()
    loss.backward()
    gth_sum = np.zeros(2)
    for r in range(R):
        term = O.PairTerm("lj", torch.tensor([1.0, 1.0]), 2.5, T(g["cell"]), p=12, q=0 if exvol else 6, c=0 if exvol else 1)
        traj, lam, gth = oracle_run(
            pos[r], g["cell"], vel[r], g["mass"][:n_atoms], [term], 1.0, 50.0, 5, t,
            lambda L: L[1][::2].pow(2).sum() / (5 * nrm) + L[0][-1].pow(2).sum() / nrm
            + (L[2][-1].sum() if nhc else 0.0), ensemble=ensemble)
        close(q_t[r], traj[1], 0, 2e-5, "q_t[%d]" % r)
        close(v_t[r], traj[0], 0, 2e-4, "v_t[%d]" % r)
        close(v0.grad[r], lam[0], 1e-3, 2e-4 * float(lam[0].abs().max()), "adj v0[%d]" % r)
        close(q0.grad[r], lam[1], 1e-3, 2e-4 * float(lam[1].abs().max()), "adj q0[%d]" % r)
        if nhc:
            close(out[2][r], traj[2], 1e-4, 1e-4, "pv_t[%d]" % r)
            close(pv0.grad[r], lam[2], 1e-3, 2e-4 * float(lam[2].abs().max()) + 1e-6, "adj pv0[%d]" % r)
        gth_sum += gth.numpy()
    got = np.array([float(mdl.sigma.grad), float(mdl.epsilon.grad)])
    close(got, gth_sum, 1e-3, 2e-4 * np.abs(gth_sum).max() + 1e-7, "dL/dtheta")


def test_ring_kernels_unwrapped_positions_vs_oracle():
    """Atoms outside the [-0.24, 1.24] cell window send the ring sweep to the clamped minimum image (the reference's
    -[s > .5] + [s < -.5], one image only): a third of the atoms shifted by a whole cell, forward + adjoint + fused
    RDF against the oracle."""
    from mdgrad_amd import ops, potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    g = load_golden("nhc_traj_lj")
    R, nT = 2, 7
    rng = np.random.default_rng(3)
    system = mk_system(g["pos"], g["cell"], g["vel"], g["mass"])
    mdl = P.LennardJones(1.0, 1.0)
    integ = NoseHooverChain(Stack({"pair": PairPotentials(system, mdl, cutoff=2.5)}), system, T=1.0, num_chains=5,
                            Q=50.0).to(DEV)
    spec = integ.fused_spec("NH_verlet")
    spec.block = 64
    pos = np.mod(g["pos"][None] + rng.normal(0, 0.03, (R,) + g["pos"].shape), g["cell"]).astype(np.float32)
    pos[:, ::3, 0] += g["cell"][0]                        # unwrapped: one cell to the right
    pos[:, 1::5, 2] -= g["cell"][2]                       # ... and one down
    vel = rng.normal(0, 1.0, pos.shape).astype(np.float32)
    t = torch.Tensor([0.005 * i for i in range(nT)])
    v0, q0 = T(vel, DEV).requires_grad_(True), T(pos, DEV).requires_grad_(True)
    pv0 = torch.zeros(R, 5, device=DEV, requires_grad=True)
    v_t, q_t, pv_t = ops.FusedTrajFn.apply(v0, q0, pv0, t.to(DEV), spec.flat_params(), spec)
    mdl.zero_grad()
    (q_t[:, ::2].pow(2).sum() / 1296 + v_t[:, -1].pow(2).sum() / 324).backward()
    gth_sum = np.zeros(2)
    for r in range(R):
        term = O.PairTerm("lj", torch.tensor([1.0, 1.0]), 2.5, T(g["cell"]), p=12, q=6, c=1)
        traj, lam, gth = oracle_run(pos[r], g["cell"], vel[r], g["mass"], [term], 1.0, 50.0, 5, t,
                                    lambda L: L[1][::2].pow(2).sum() / 1296 + L[0][-1].pow(2).sum() / 324)
        close(q_t[r], traj[1], 0, 4e-5, "q_t[%d]" % r)
        close(v0.grad[r], lam[0], 1e-3, 2e-4 * float(lam[0].abs().max()), "adj v0[%d]" % r)
        close(q0.grad[r], lam[1], 1e-3, 2e-4 * float(lam[1].abs().max()), "adj q0[%d]" % r)
        gth_sum += gth.numpy()
    got = np.array([float(mdl.sigma.grad), float(mdl.epsilon.grad)])
    close(got, gth_sum, 1e-3, 2e-4 * np.abs(gth_sum).max(), "dL/dtheta")


@pytest.mark.parametrize("chains", [2, 3, 16])
def test_ring_kernels_chain_lengths_vs_oracle(chains):
    """The thermostat chain lives one entry per lane in the ring kernels (DPP row shifts between neighbours): the
    shortest and the longest chain the kernels take, against the oracle."""
    from mdgrad_amd import ops, potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    g = load_golden("nhc_traj_lj")
    R, nT = 2, 9
    rng = np.random.default_rng(chains)
    system = mk_system(g["pos"], g["cell"], g["vel"], g["mass"])
    mdl = P.LennardJones(1.0, 1.0)
    integ = NoseHooverChain(Stack({"pair": PairPotentials(system, mdl, cutoff=2.5)}), system, T=1.0, num_chains=chains,
                            Q=50.0).to(DEV)
    spec = integ.fused_spec("NH_verlet")
    spec.block = 64
    pos = np.mod(g["pos"][None] + rng.normal(0, 0.03, (R,) + g["pos"].shape), g["cell"]).astype(np.float32)
    vel = rng.normal(0, 1.0, pos.shape).astype(np.float32)
    t = torch.Tensor([0.005 * i for i in range(nT)])
    v0, q0 = T(vel, DEV).requires_grad_(True), T(pos, DEV).requires_grad_(True)
    pv0 = torch.zeros(R, chains, device=DEV, requires_grad=True)
    v_t, q_t, pv_t = ops.FusedTrajFn.apply(v0, q0, pv0, t.to(DEV), spec.flat_params(), spec)
    mdl.zero_grad()
    (q_t[:, ::2].pow(2).sum() / 1620 + v_t[:, -1].pow(2).sum() / 324 + pv_t[:, -1].sum()).backward()
    gth_sum = np.zeros(2)
    for r in range(R):
        term = O.PairTerm("lj", torch.tensor([1.0, 1.0]), 2.5, T(g["cell"]), p=12, q=6, c=1)
        traj, lam, gth = oracle_run(pos[r], g["cell"], vel[r], g["mass"], [term], 1.0, 50.0, chains, t,
                                    lambda L: L[1][::2].pow(2).sum() / 1620 + L[0][-1].pow(2).sum() / 324 + L[2][-1].sum())
        close(q_t[r], traj[1], 0, 2e-5, "q_t[%d]" % r)
        close(pv_t[r], traj[2], 1e-4, 1e-4, "pv_t[%d]" % r)
        close(v0.grad[r], lam[0], 1e-3, 2e-4 * float(lam[0].abs().max()), "adj v0[%d]" % r)
        close(q0.grad[r], lam[1], 1e-3, 2e-4 * float(lam[1].abs().max()), "adj q0[%d]" % r)
        close(pv0.grad[r], lam[2], 1e-3, 2e-4 * float(lam[2].abs().max()) + 1e-6, "adj pv0[%d]" % r)
        gth_sum += gth.numpy()
    got = np.array([float(mdl.sigma.grad), float(mdl.epsilon.grad)])
    close(got, gth_sum, 1e-3, 2e-4 * np.abs(gth_sum).max(), "dL/dtheta")


@pytest.mark.parametrize("name,ensemble", [("ljfam_8_4", "nhc"), ("lj69", "nve"), ("exvol10", "nhc"), ("morse_pos", "nhc"),
                                           ("morse_neg", "nve"), ("buck", "nhc"), ("yukawa", "nhc")])
def test_ring_kernels_other_pair_forms_vs_oracle(name, ensemble):
    """Every built-in single-term form on the wave-per-replica kernels (through pair_eval instead of the LJ 12-6
    polynomial): trajectory, adjoints and parameter gradients of 3 replicas == oracle."""
    from mdgrad_amd import ops, potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NVE, NoseHooverChain
    from test_gpu_parity import form
    g = load_golden("nhc_traj_lj")
    R, nT, n_atoms = 3, 7, 108
    rng = np.random.default_rng(len(name))
    system = mk_system(g["pos"], g["cell"], g["vel"], g["mass"])
    oracle_of = {"ljfam_8_4": ("lj", [0.95, 1.1], dict(p=8, q=4, c=1)), "lj69": ("lj", [1.0, 1.2], dict(p=9, q=6, c=1)),
                 "exvol10": ("lj", [1.1, 0.7], dict(p=10, q=0, c=0)), "morse_pos": ("morse", [], dict(a=3.0, phi=1.5)),
                 "morse_neg": ("morse", [], dict(a=2.5, phi=-1.2)), "buck": ("buck", [1000.0, 3.5, 5.0], {}),
                 "yukawa": ("yukawa", [1.3, 0.8], {})}[name]
    mdl = P.Yukawa(epsilon=1.3, kappa=0.8) if name == "yukawa" else form(name)
    stack = Stack({"pair": PairPotentials(system, mdl, cutoff=2.5)})
    nhc = ensemble == "nhc"
    integ = (NoseHooverChain(stack, system, T=1.0, num_chains=5, Q=50.0) if nhc else NVE(stack, system)).to(DEV)
    spec = integ.fused_spec("NH_verlet" if nhc else "verlet")
    spec.block = 64
    pos = np.mod(g["pos"][None] + rng.normal(0, 0.02, (R,) + g["pos"].shape), g["cell"]).astype(np.float32)
    vel = rng.normal(0, 0.5, pos.shape).astype(np.float32)
    t = torch.Tensor([0.004 * i for i in range(nT)])
    v0, q0 = T(vel, DEV).requires_grad_(True), T(pos, DEV).requires_grad_(True)
    pv0 = torch.zeros(R, 5, device=DEV, requires_grad=True) if nhc else None
    out = ops.FusedTrajFn.apply(v0, q0, pv0, t.to(DEV), spec.flat_params(), spec)
    v_t, q_t = out[0], out[1]
    mdl.zero_grad()
    nrm = n_atoms * 3
    loss = (q_t[:, ::2].pow(2).sum((1, 2, 3)) / (4 * nrm) + v_t[:, -1].pow(2).sum((1, 2)) / nrm).sum()
    if nhc:
        loss = loss + out[2][:, -1].sum()
    loss.backward()
    params = list(mdl.parameters())
    gth_sum = np.zeros(len(params))
    for r in range(R):
        term = O.PairTerm(oracle_of[0], torch.tensor(oracle_of[1]), 2.5, T(g["cell"]), **oracle_of[2])
        traj, lam, gth = oracle_run(
            pos[r], g["cell"], vel[r], g["mass"], [term], 1.0, 50.0, 5, t,
            lambda L: L[1][::2].pow(2).sum() / (4 * nrm) + L[0][-1].pow(2).sum() / nrm + (L[2][-1].sum() if nhc else 0.0),
            ensemble=ensemble)
        close(q_t[r], traj[1], 0, 3e-5, "q_t[%d]" % r)
        close(v_t[r], traj[0], 0, 3e-4, "v_t[%d]" % r)
        close(v0.grad[r], lam[0], 2e-3, 3e-4 * float(lam[0].abs().max()), "adj v0[%d]" % r)
        close(q0.grad[r], lam[1], 2e-3, 3e-4 * float(lam[1].abs().max()), "adj q0[%d]" % r)
        if params:
            gth_sum += gth.numpy()
    if params:
        got = np.array([float(p.grad) for p in params])
        close(got, gth_sum, 2e-3, 3e-4 * np.abs(gth_sum).max() + 1e-7, "dL/dtheta")


# ------------------------------------------------------------------ many-frame RDF kernels
def _oracle_rdf_chunked(frames, cell, nbins, r_range, width, wgt, chunk=100):
    """g(r) and d(sum g wgt)/dxyz from the oracle, the raw histogram accumulated over chunks of frames."""
    raws = []
    for k in range(0, frames.shape[0], chunk):
        with torch.no_grad():
            raws.append(O.rdf_raw_oracle(T(frames[k:k + chunk]), T(cell), nbins, r_range, width=width))
    raw = torch.stack(raws).sum(0).requires_grad_(True)
    _, _, g = O.rdf_normalise_oracle(raw, nbins, r_range)
    (g_raw,) = torch.autograd.grad((g * wgt).sum(), raw)
    grads = []
    for k in range(0, frames.shape[0], chunk):
        x = T(frames[k:k + chunk]).requires_grad_(True)
        (gx,) = torch.autograd.grad((O.rdf_raw_oracle(x, T(cell), nbins, r_range, width=width) * g_raw).sum(), x)
        grads.append(gx)
    return g.detach(), torch.cat(grads)


@pytest.mark.parametrize("width_scale", [1.0, 1.6, 0.4])
def test_many_frame_rdf_kernels_vs_oracle(width_scale):
    """1 100 frames of the 108-atom box through rdf(): half-width forward + fine-table backward (width =
    spacing), full-width forward + recurrence backward (1.6 x), narrow full-width (0.4 x) -- g(r) and
    d(sum g w)/dxyz against the oracle itself (not a sibling kernel)."""
    from mdgrad_amd.observable import rdf
    g = load_golden("rdf")
    rng = np.random.default_rng(11)
    base = g["xyz"][0]
    frames = np.stack([np.mod(base + rng.normal(0, 0.05, base.shape), g["cell"]) for _ in range(1100)]).astype(np.float32)
    nbins, rr = 100, (0.75, 2.5)
    spacing = (rr[1] - rr[0]) / (nbins - 1)
    width = None if width_scale == 1.0 else width_scale * spacing
    wgt = torch.linspace(-1, 1, nbins)
    system = mk_system(base, g["cell"])
    x = T(frames, DEV).requires_grad_(True)
    count, bins, gr = rdf(system, nbins=nbins, r_range=rr, width=width)(x)
    (gx,) = torch.autograd.grad((gr * wgt.to(DEV)).sum(), x)
    g_o, gx_o = _oracle_rdf_chunked(frames, g["cell"], nbins, rr, width, wgt)
    close(gr, g_o, 1e-4, 1e-4, "g(r), 1100 frames, width x%.1f" % width_scale)            # g(r): abs 1e-4
    close(gx, gx_o, 1e-3, 1e-4 * float(gx_o.abs().max()), "d(g.w)/dxyz, 1100 frames, width x%.1f" % width_scale)


# ------------------------------------------------------------------ large-N fused path vs the oracle
def _large_case_nve(n_side, n_frames, seed):
    """The NVE branch of the multi-launch kernels (md.py:133-150, sovlers.py:42-101) against the oracle."""
    from mdgrad_amd import potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NVE
    from mdgrad_amd.sovlers import odeint_adjoint
    pos, cell = liquid(n_side, seed=seed, jitter=0.05)
    rng = np.random.default_rng(seed + 100)
    vel = rng.normal(0, 1.0, pos.shape).astype(np.float32)
    mass = np.full(len(pos), 1.008, dtype=np.float32)
    t = torch.Tensor([0.004 * i for i in range(n_frames)])
    system = mk_system(pos, cell, vel, mass)
    mdl = P.LennardJones(1.0, 1.0)
    integ = NVE(Stack({"p": PairPotentials(system, mdl, cutoff=2.5)}), system).to(DEV)
    integ.fused_large = True
    assert integ.fused_spec("verlet").large
    y0 = [s.clone().requires_grad_(True) for s in integ.get_inital_states(wrap=True)]
    v_t, q_t = odeint_adjoint(integ, tuple(y0), t.to(DEV), method="verlet")
    term = O.PairTerm("lj", torch.tensor([1.0, 1.0]), 2.5, T(cell), p=12, q=6, c=1)
    N = len(pos)

    def loss_fn(L):
        return L[1][::2].pow(2).sum() / (L[1][::2].numel()) + L[0][-1].pow(2).sum() / (N * 3)

    loss_fn((v_t, q_t)).backward()
    traj, lam, gth = oracle_run(pos, cell, vel, mass, [term], 1.0, 30.0, 3, t, loss_fn, ensemble="nve")
    close(q_t, traj[1], 0, 2e-5, "q_t (NVE, N=%d)" % N)
    close(v_t, traj[0], 0, 5e-4, "v_t (NVE, N=%d)" % N)
    got = torch.stack([mdl.sigma.grad.reshape(()), mdl.epsilon.grad.reshape(())])
    close(got, gth, 2e-3, 2e-4 * float(gth.abs().max()), "dL/dtheta (NVE, N=%d)" % N)
    for y, l, nm in zip(y0, lam, ("adj v0", "adj q0")):
        close(y.grad, l, 2e-3, 5e-4 * float(l.abs().max()) + 1e-9, "%s (NVE, N=%d)" % (nm, N))


def _large_case(n_side, n_frames, adjoint, seed):
    from mdgrad_amd import potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    from mdgrad_amd.sovlers import odeint_adjoint
    pos, cell = liquid(n_side, seed=seed, jitter=0.05)
    rng = np.random.default_rng(seed + 100)
    vel = rng.normal(0, 1.0, pos.shape).astype(np.float32)
    mass = np.full(len(pos), 1.008, dtype=np.float32)
    t = torch.Tensor([0.004 * i for i in range(n_frames)])
    system = mk_system(pos, cell, vel, mass)
    mdl = P.LennardJones(1.0, 1.0)
    integ = NoseHooverChain(Stack({"p": PairPotentials(system, mdl, cutoff=2.5)}), system, T=1.0, num_chains=3,
                            Q=30.0).to(DEV)
    integ.fused_large = True
    assert integ.fused_spec("NH_verlet").large
    y0 = [s.clone().requires_grad_(True) for s in integ.get_inital_states(wrap=True)]
    v_t, q_t, pv_t = odeint_adjoint(integ, tuple(y0), t.to(DEV), method="NH_verlet")
    term = O.PairTerm("lj", torch.tensor([1.0, 1.0]), 2.5, T(cell), p=12, q=6, c=1)
    N = len(pos)

    def loss_fn(L):
        return L[1][::2].pow(2).sum() / (L[1][::2].numel()) + L[0][-1].pow(2).sum() / (N * 3) + L[2][-1].sum() * 1e-3

    if adjoint:
        loss_fn((v_t, q_t, pv_t)).backward()
        traj, lam, gth = oracle_run(pos, cell, vel, mass, [term], 1.0, 30.0, 3, t, loss_fn)
    else:
        eom = O.NHCOracle(O.ModelOracle([term]), T(mass), 1.0, 30.0, 3)
        traj = O.odeint_oracle(eom, (T(vel), T(pos), torch.zeros(3)), t)
    close(q_t, traj[1], 0, 2e-5, "q_t (N=%d)" % N)
    close(v_t, traj[0], 0, 5e-4, "v_t (N=%d)" % N)
    close(pv_t, traj[2], 1e-3, 1e-4, "pv_t (N=%d)" % N)
    if adjoint:
        got = torch.stack([mdl.sigma.grad.reshape(()), mdl.epsilon.grad.reshape(())])
        close(got, gth, 2e-3, 2e-4 * float(gth.abs().max()), "dL/dtheta (N=%d)" % N)
        for y, l, nm in zip(y0, lam, ("adj v0", "adj q0", "adj pv0")):
            close(y.grad, l, 2e-3, 5e-4 * float(l.abs().max()) + 1e-9, "%s (N=%d)" % (nm, N))


@pytest.mark.parametrize("n_side", [10, 14])
def test_large_path_vs_oracle_forward_and_adjoint(n_side):
    """traj_large at N = 1 000 and N = 2 744 (tile staging, multi-block partial reduction, neighbour buffer
    all exercised), 3 steps forward + adjoint, against the oracle."""
    _large_case(n_side, 4, True, seed=20 + n_side)


@pytest.mark.parametrize("n_side", [5, 11])
def test_large_path_nve_vs_oracle_forward_and_adjoint(n_side):
    """NVE above one workgroup's reach: 125 atoms forced onto the multi-launch kernels (all-atom scan) and 1 331 atoms
    (cell-binned scan), 4 frames forward + adjoint."""
    _large_case_nve(n_side, 4, seed=50 + n_side)


@pytest.mark.parametrize("case", ["reuse", "moved", "crowded"])
def test_large_path_stored_lists_equal_fresh_searches(case):
    """The adjoint of the multi-launch kernels re-tests the forward pass's stored candidates (4 % skin) instead of
    searching again; `block = -1` searches at every evaluation.  Same pair sets, so the same numbers up to summation
    order -- also when the midpoint moves farther than skin / 2 (time step 0.06: the device-side check sends that
    evaluation back to a search) and when an atom has more candidates than a stored row holds (a crowded cluster: the
    frame is marked and searched)."""
    from mdgrad_amd import ops, potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    pos, cell = liquid(9, seed=77, jitter=0.05)
    rng = np.random.default_rng(5)
    vscale = 1.0
    if case == "crowded":
        # a 5 x 5 x 5 cluster at spacing 0.5 in place of the liquid there: its inner atoms have > 96 candidates
        g5 = np.stack(np.meshgrid(*[np.arange(5)] * 3, indexing="ij"), -1).reshape(-1, 3) * 0.5 + 1.25
        keep = ~np.all((pos > 0.9) & (pos < 3.6), axis=1)
        pos = np.concatenate([g5.astype(np.float32), pos[keep]])
    if case == "moved":
        vscale = 8.0                                       # |v| h / 2 ~ 0.08 > skin / 2 = 0.05 for many atoms
    vel = (vscale * rng.normal(0, 1.0, pos.shape)).astype(np.float32)
    mass = np.full(len(pos), 1.008, dtype=np.float32)
    dt = 0.02 if case == "moved" else 0.003
    t = torch.Tensor([dt * i for i in range(4)]).to(DEV)
    system = mk_system(pos, cell, vel, mass)
    # (a soft, weak repulsion keeps these contrived states finite; the test is about the pair sets, not the physics)
    mdl = (P.LennardJones(1.0, 1.0) if case == "reuse" else
           P.ExcludedVolume(0.3, 1e-3, 12) if case == "crowded" else P.ExcludedVolume(0.8, 1e-5, 12))
    from mdgrad_amd.md import NVE
    stack = Stack({"p": PairPotentials(system, mdl, cutoff=2.5)})
    nhc = case != "moved"                                  # (fast atoms would blow the thermostat up: NVE there)
    integ = (NoseHooverChain(stack, system, T=1.0, num_chains=3, Q=30.0) if nhc else NVE(stack, system)).to(DEV)
    integ.fused_large = True
    res = []
    stats0 = dict(ops.LARGE_STATS)
    for block in (0, -1):
        spec = integ.fused_spec("NH_verlet" if nhc else "verlet")
        spec.block = block
        y0 = [s.clone().requires_grad_(True) for s in integ.get_inital_states(wrap=True)]
        out = ops.FusedTrajFn.apply(y0[0], y0[1], y0[2] if nhc else None, t, spec.flat_params(), spec)
        v_t, q_t = out[0], out[1]
        mdl.zero_grad()
        (q_t[::2].pow(2).mean() + v_t[-1].pow(2).mean() + (out[2][-1].sum() * 1e-3 if nhc else 0.0)).backward()
        res.append([q_t.detach(), v_t.detach(), y0[0].grad.clone(), y0[1].grad.clone(),
                    torch.stack([p.grad.reshape(()) for p in mdl.parameters()])])
    assert ops.LARGE_STATS["adjoint_redone_with_searches"] - stats0["adjoint_redone_with_searches"] == (case == "moved")
    assert ops.LARGE_STATS["lists_incomplete"] - stats0["lists_incomplete"] == (case == "crowded")
    # (the forward pass searches with the skin when it keeps the lists: extra candidates beyond the cutoff contribute
    #  nothing but change the lane assignment of the sums)
    for a, b, nm in zip(res[0], res[1], ("q_t", "v_t", "adj v0", "adj q0", "dL/dtheta")):
        close(a, b, 1e-4, 1e-5 * float(b.abs().max()) + 1e-12, "%s (%s)" % (nm, case))


def test_large_path_lists_reused_across_forward_steps_vs_oracle_and_fresh_searches():
    """Verlet reuse in the forward pass: 1 000 atoms, 17 frames at dt 0.005 with thermal velocities -- the first list
    serves several steps, an atom then leaves its 0.45-skin ball and the device asks for a new search (more than one
    build, far fewer than frames; the last frame keeps the wider margin its adjoint midpoint needs).  Trajectory,
    adjoints and dL/dtheta against the oracle (exact neighbour list at every evaluation), and against the same
    kernels searching at every evaluation (block = -1)."""
    from mdgrad_amd import ops, potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    pos, cell = liquid(10, seed=31, jitter=0.05)
    rng = np.random.default_rng(131)
    vel = rng.normal(0, 1.0, pos.shape).astype(np.float32)
    mass = np.full(len(pos), 1.008, dtype=np.float32)
    n_frames = 17
    t = torch.Tensor([0.005 * i for i in range(n_frames)])
    system = mk_system(pos, cell, vel, mass)
    mdl = P.LennardJones(1.0, 1.0)
    integ = NoseHooverChain(Stack({"p": PairPotentials(system, mdl, cutoff=2.5)}), system, T=1.0, num_chains=3,
                            Q=30.0).to(DEV)
    integ.fused_large = True
    N = len(pos)

    def loss_fn(L):
        return L[1][::4].pow(2).sum() / (L[1][::4].numel()) + L[0][-1].pow(2).sum() / (N * 3) + L[2][-1].sum() * 1e-3

    res = []
    stats0 = dict(ops.LARGE_STATS)
    for block in (0, -1):
        spec = integ.fused_spec("NH_verlet")
        assert spec.large
        spec.block = block
        y0 = [s_.clone().requires_grad_(True) for s_ in integ.get_inital_states(wrap=True)]
        v_t, q_t, pv_t = ops.FusedTrajFn.apply(y0[0], y0[1], y0[2], t.to(DEV), spec.flat_params(), spec)
        if block == 0:
            builds = ops.large_list_builds(spec)[0].tolist()
            assert builds[0] == 0 and all(b <= f for f, b in enumerate(builds)) and builds == sorted(builds)
            assert 2 <= len(set(builds)) <= n_frames // 2, builds          # reused, and rebuilt at least once
        mdl.zero_grad()
        loss_fn((v_t, q_t, pv_t)).backward()
        res.append([q_t.detach(), v_t.detach(), pv_t.detach(), y0[0].grad.clone(), y0[1].grad.clone(), y0[2].grad.clone(),
                    torch.stack([mdl.sigma.grad.reshape(()), mdl.epsilon.grad.reshape(())])])
    assert ops.LARGE_STATS["adjoint_redone_with_searches"] == stats0["adjoint_redone_with_searches"]
    assert ops.LARGE_STATS["lists_incomplete"] == stats0["lists_incomplete"]
    names = ("q_t", "v_t", "pv_t", "adj v0", "adj q0", "adj pv0", "dL/dtheta")
    for a, b, nm in zip(res[0], res[1], names):
        close(a, b, 2e-4, 2e-5 * float(b.abs().max()) + 1e-12, "%s (lists vs searches)" % nm)
    term = O.PairTerm("lj", torch.tensor([1.0, 1.0]), 2.5, T(cell), p=12, q=6, c=1)
    traj, lam, gth = oracle_run(pos, cell, vel, mass, [term], 1.0, 30.0, 3, t, loss_fn)
    close(res[0][0], traj[1], 0, 1e-4, "q_t vs oracle")
    close(res[0][1], traj[0], 0, 2e-3, "v_t vs oracle")
    close(res[0][2], traj[2], 2e-3, 5e-4, "pv_t vs oracle")
    close(res[0][6], gth, 5e-3, 5e-4 * float(gth.abs().max()), "dL/dtheta vs oracle")
    for g, l, nm in zip(res[0][3:6], lam, ("adj v0", "adj q0", "adj pv0")):
        close(g, l, 5e-3, 1e-3 * float(l.abs().max()) + 1e-9, "%s vs oracle" % nm)


def test_large_path_nve_two_replicas_lists_reused_equal_fresh_searches():
    """NVE, two stacked replicas of 1 331 atoms with different velocities (their lists are rebuilt at different steps:
    the rebuild decision is per replica, on the device), 21 frames: stored lists reused across steps == a search at
    every evaluation, forward and adjoint."""
    from mdgrad_amd import ops, potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NVE
    pos, cell = liquid(11, seed=41, jitter=0.05)
    rng = np.random.default_rng(141)
    N = len(pos)
    mass = np.full(N, 1.008, dtype=np.float32)
    system = mk_system(pos, cell, rng.normal(0, 1.0, pos.shape).astype(np.float32), mass)
    mdl = P.LennardJones(1.0, 1.0)
    integ = NVE(Stack({"p": PairPotentials(system, mdl, cutoff=2.5)}), system).to(DEV)
    integ.fused_large = True
    t = torch.Tensor([0.005 * i for i in range(21)]).to(DEV)
    q0 = T(np.stack([pos, np.mod(pos + rng.normal(0, 0.02, pos.shape), cell)]).astype(np.float32), DEV)
    v0 = T(np.stack([rng.normal(0, 0.7, pos.shape), rng.normal(0, 1.6, pos.shape)]).astype(np.float32), DEV)
    res = []
    for block in (0, -1):
        spec = integ.fused_spec("verlet")
        assert spec.large
        spec.block = block
        v, q = v0.clone().requires_grad_(True), q0.clone().requires_grad_(True)
        v_t, q_t = ops.FusedTrajFn.apply(v, q, None, t, spec.flat_params(), spec)
        if block == 0:
            builds = ops.large_list_builds(spec).tolist()
            assert all(2 <= len(set(b)) <= 12 for b in builds), builds
            assert len(set(builds[1])) > len(set(builds[0])), "the hotter replica searches more often: %r" % (builds,)
        mdl.zero_grad()
        (q_t[:, ::4].pow(2).mean() + v_t[:, -1].pow(2).mean()).backward()
        res.append([q_t.detach(), v_t.detach(), v.grad.clone(), q.grad.clone(),
                    torch.stack([mdl.sigma.grad.reshape(()), mdl.epsilon.grad.reshape(())])])
    for a, b, nm in zip(res[0], res[1], ("q_t", "v_t", "adj v0", "adj q0", "dL/dtheta")):
        close(a, b, 2e-4, 2e-5 * float(b.abs().max()) + 1e-12, "%s (NVE, lists vs searches)" % nm)


def test_three_bins_per_side_large_path_and_cell_rdf_vs_oracle():
    """The smallest box the cell-binned kernels accept: exactly three bins per side (512-atom LJ liquid, L = 8.46 =
    3 x 2.82), where every stencil wraps around and covers the whole box.  The multi-launch trajectory kernels
    (row-based search, listed kernels) forward + adjoint, and the list-free RDF sweeps (half stencil forward, full
    stencil backward), each against the oracle."""
    from mdgrad_amd import ops
    from mdgrad_amd.observable import rdf
    pos, cell = liquid(8, seed=61, jitter=0.08)
    assert 3 * 2.8 <= float(cell[0]) < 4 * 2.8
    _large_case(8, 9, True, seed=61)
    rng = np.random.default_rng(62)
    frames = np.stack([np.mod(pos + rng.normal(0, 0.06, pos.shape), cell) for _ in range(4)]).astype(np.float32)
    frames[2, ::5] -= np.asarray(cell, dtype=np.float32)                  # unwrapped coordinates
    system = mk_system(pos, cell)
    wgt = torch.linspace(-1, 1, 100)
    x = T(frames, DEV).requires_grad_(True)
    was = ops.RDF_LIST_ATOMS
    ops.RDF_LIST_ATOMS = 256                                          # (the cell path normally starts at 2 048 atoms)
    try:
        obs = rdf(system, nbins=100, r_range=(0.75, 2.5))
        count, bins, gr = obs(x)
        (gx,) = torch.autograd.grad((gr * wgt.to(DEV)).sum(), x)
    finally:
        ops.RDF_LIST_ATOMS = was
    xo = T(frames).requires_grad_(True)
    _, _, go = O.rdf_oracle(xo, T(cell), 100, (0.75, 2.5))
    (gxo,) = torch.autograd.grad((go * wgt).sum(), xo)
    close(gr, go, 1e-4, 1e-4, "g(r), three bins per side")
    close(gx, gxo, 1e-3, 1e-4 * float(gxo.abs().max()), "d(g.w)/dxyz, three bins per side")


def test_large_path_4096_atoms_one_step_vs_oracle():
    """BASELINE config #4's size: one forward NH-Verlet step of the 4 096-atom LJ liquid against the oracle."""
    _large_case(16, 2, False, seed=36)


# ------------------------------------------------------------------ BASELINE config #3 as a golden
def test_config3_water192_schnet_trajectory_adjoint_golden():
    """192-atom all-atom water box, SchNet A128/F128/G32/3 conv + ExcludedVolume prior, NHC, O-H RDF loss:
    trajectory, g(r), 185 571 parameter gradients and the adjoint w.r.t. the initial state vs the
    reference (golden G14)."""
    from mdgrad_amd import potentials as P
    from mdgrad_amd.interface import GNNPotentials, PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    from mdgrad_amd.nn import get_model
    from mdgrad_amd.observable import rdf
    from mdgrad_amd.sovlers import odeint_adjoint
    from test_gpu_schnet import params_of, sd_of
    g = load_golden("gnn_traj_water192")
    system = mk_system(g["pos"], g["cell"], g["vel"], g["masses"], g["numbers"])
    net = get_model(params_of(g))
    net.load_state_dict(sd_of(g))
    gnn = GNNPotentials(system, net, cutoff=float(g["cutoff"]))
    prior = PairPotentials(system, P.ExcludedVolume(float(g["prior_sigma"]), float(g["prior_epsilon"]), 12),
                           cutoff=float(g["cutoff"]))
    integ = NoseHooverChain(Stack({"gnn": gnn, "prior": prior}), system, T=float(g["T"]),
                            num_chains=int(g["chains"]), Q=float(g["Q"]), adjoint=True).to(DEV)
    assert [n for n, _ in integ.named_parameters()] == [str(x) for x in g["param_names"]]
    y0 = [s.clone().requires_grad_(True) for s in integ.get_inital_states(wrap=True)]
    t = torch.Tensor([float(g["dt"]) * i for i in range(g["q_t"].shape[0])]).to(DEV)
    v_t, q_t, pv_t = odeint_adjoint(integ, tuple(y0), t, method="NH_verlet")
    for x, k in zip((v_t, q_t, pv_t), ["v_t", "q_t", "pv_t"]):
        close(x, g[k], 1e-4, 1e-4 * max(1e-3, np.abs(g[k]).max()), k)
    obs = rdf(system, nbins=40, r_range=(0.6, 5.0), index_tuple=(g["idx_O"].tolist(), g["idx_H"].tolist()))
    _, _, gr = obs(q_t[::2])
    close(gr, g["g"], 1e-3, 2e-4, "g_OH")
    loss = gr.pow(2).mean() + q_t[-1].pow(2).mean() * 1e-3 + v_t[-1].pow(2).sum() * 1e-2
    close(loss.reshape(1), g["loss"], 1e-4, 1e-6, "loss")
    loss.backward()
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in integ.parameters()])
    close(flat, g["grad_flat"], 5e-3, 2e-4 * np.abs(g["grad_flat"]).max(), "dL/dtheta (185 571 params)")
    close(y0[1].grad, g["grad_q0"], 5e-3, 2e-3 * np.abs(g["grad_q0"]).max(), "grad_q0")
    close(y0[0].grad, g["grad_v0"], 5e-3, 2e-3 * np.abs(g["grad_v0"]).max(), "grad_v0")


# ------------------------------------------------------------------ neighbour-list capacity growth
def test_build_ell_grows_when_a_cluster_exceeds_the_density_estimate():
    """A dilute box with one dense droplet: the longest row is far above 1.5 x the mean density, so the first
    pass overflows and the list is rebuilt with the capacity the builder reported; the result equals the oracle."""
    from mdgrad_amd import ops, _lib
    rng = np.random.default_rng(3)
    L = 30.0
    gas = rng.uniform(0, L, (300, 3))
    drop = 15.0 + rng.normal(0, 0.9, (120, 3))
    pos = np.concatenate([gas, drop]).astype(np.float32)
    cell = np.array([L, L, L], dtype=np.float32)
    cs = _lib.make_cell(cell)
    est = ops.estimate_max_nbr(len(pos), cs, 2.5)
    ell = ops.build_ell(T(pos, DEV), cs, 2.5)
    longest = int(ell.cnt.max())
    assert longest > est, "the droplet must overflow the estimate for this test to mean anything"
    assert ell.max_nbr >= longest
    nbr, off = ell.half_list()
    onbr, ooff = O.nbr_list(T(pos), 2.5, T(cell))
    assert np.array_equal(nbr.cpu().numpy(), onbr.numpy()) and np.array_equal(off.cpu().numpy(), ooff.numpy())


def test_cell_list_for_replica_stacked_systems_equals_dense():
    """Grouped (System.replicate) neighbour search through the cell list: every replica bins on its own; the list
    is identical to the dense group-restricted search, with and without an exclusion mask."""
    from mdgrad_amd import ops, _lib
    R = 3
    pos, cell = liquid(10, seed=4)
    rng = np.random.default_rng(8)
    stack = np.concatenate([np.mod(pos + rng.normal(0, 0.1, pos.shape), cell) for _ in range(R)]).astype(np.float32)
    stack[5::13] += cell[0]                               # unwrapped atoms bin correctly too
    cs = _lib.make_cell(cell)
    x = T(stack, DEV)
    n = len(pos)
    for mask in (None, ops.build_mask(n, ex_pairs=[[0, 1], [7, 400], [10, 11]], device=DEV)):
        a = ops.build_ell(x, cs, 2.5, mask=mask, method="dense", group=n)
        b = ops.build_ell(x, cs, 2.5, mask=mask, method="cell", group=n, max_nbr=a.max_nbr)
        assert torch.equal(a.cnt, b.cnt)
        k = torch.arange(a.max_nbr, device=DEV)[None, :] < a.cnt[:, None]
        assert torch.equal(a.col[k], b.col[k]) and torch.equal(a.shift[k], b.shift[k])
        assert int((a.col[k] // n != torch.arange(R * n, device=DEV)[:, None].expand_as(a.col)[k] // n).sum()) == 0


def test_vacf_and_temperature_kernels_golden():
    """Golden G16 (the reference's vacf / Temperature incl. the gradient of a weighted sum w.r.t. the velocities)."""
    from mdgrad_amd.observable import vacf
    from mdgrad_amd.thermo import Temperature
    g = load_golden("vacf_temp")
    system = mk_system(g["pos"], g["cell"], g["vel"], g["masses"])
    v = T(g["v_t"], DEV).requires_grad_(True)
    c = vacf(system, t_range=12)(v)
    close(c, g["vacf"], 1e-5, 1e-7, "vacf")
    (gv,) = torch.autograd.grad((c * T(g["wgt"], DEV)).sum(), v)
    close(gv, g["vacf_grad"], 1e-5, 1e-7 * np.abs(g["vacf_grad"]).max() + 1e-10, "d vacf / dv")
    temp = Temperature(system)
    v1 = T(g["vel"], DEV).requires_grad_(True)
    T1 = temp(v1)
    assert T1.dim() == 0
    close(T1.reshape(1), g["T_single"], 2e-6, 0, "T")
    (gT,) = torch.autograd.grad(T1, v1)
    close(gT, g["T_grad"], 2e-6, 1e-9, "dT/dv")
    close(temp(T(g["v_t"], DEV)), g["T_frames"], 2e-6, 0, "T per frame")
    assert torch.equal(vacf(system, t_range=12)(v), c), "bitwise reproducible"


def test_fused_adjoint_returns_time_vjps_like_the_generic_path():
    """dL/dt (`time_vjps`, sovlers.py:258-266,293): the fused trajectory op returns it when t requires grad --
    compared with the generic path (the reference's control flow on the HIP autograd ops) and with the closed form
    f(y_k) . dL/dy_k built from the oracle's right-hand side."""
    from mdgrad_amd.sovlers import odeint_adjoint, OdeintAdjointMethod
    from mdgrad_amd.tinydiffeq import _flatten
    g = load_golden("nhc_traj_lj")
    res = []
    for fused in (True, False):
        system, mdl, integ = lj_setup(g)
        y0 = tuple(integ.get_inital_states(wrap=True))
        t = torch.Tensor([0.005 * i for i in range(8)]).to(DEV).requires_grad_(True)
        if fused:
            out = odeint_adjoint(integ, y0, t, method="NH_verlet")
        else:
            out = OdeintAdjointMethod.apply(*y0, integ, t, _flatten(integ.parameters()), 1e-6, 1e-12, "NH_verlet", None)
        (out[1].pow(2).mean() + out[0][-1].pow(2).mean() + out[2][-1].sum()).backward()
        res.append((t.grad.clone(), [o.detach() for o in out]))
    close(res[0][0], res[1][0], 1e-4, 1e-5 * float(res[1][0].abs().max()), "time_vjps fused vs generic")
    assert abs(float(res[0][0].sum())) <= 1e-4 * float(res[0][0].abs().max()), "dL/dt_0 = -sum of the others"
    # closed form from the oracle's RHS at the saved frames
    term = O.PairTerm("lj", torch.tensor([1.0, 1.0]), 2.5, T(g["cell"]), p=12, q=6, c=1)
    eom = O.NHCOracle(O.ModelOracle([term]), T(g["mass"]), 1.0, 50.0, 5)
    v_t, q_t, pv_t = [o.cpu() for o in res[0][1]]
    n = q_t.shape[0]
    for k in (1, n - 1):
        a, vv, b = eom.rhs((v_t[k], q_t[k], pv_t[k]))
        gq = 2 * q_t[k] / q_t.numel()
        expect = float((vv * gq).sum())
        if k == n - 1:
            expect += float((a * 2 * v_t[k] / v_t[k].numel()).sum() + b.sum())
        assert abs(float(res[0][0][k]) - expect) <= 1e-3 * abs(expect) + 1e-6, "time_vjps[%d]" % k


def test_fit_loop_parameter_trajectory_matches_an_oracle_run_loop():
    """the outer training loop (simulate -> rdf -> loss -> adjoint -> Adam) run on the HIP
    path and, with the same inputs, on the CPU oracle: losses and the (sigma, epsilon) trajectory over 4 optimizer
    steps agree (2 replicas x 11 MD steps per epoch, pooled RDF, target g = 1)."""
    from mdgrad_amd import ops
    from mdgrad_amd import potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    from mdgrad_amd.observable import rdf
    g = load_golden("nhc_traj_lj")
    R, nT, epochs, lr = 2, 12, 4, 0.02
    rng = np.random.default_rng(21)
    pos = np.stack([np.mod(g["pos"] + rng.normal(0, 0.03, g["pos"].shape), g["cell"]) for _ in range(R)]).astype(np.float32)
    vel = rng.normal(0, 1.0, pos.shape).astype(np.float32)
    t = torch.Tensor([0.005 * i for i in range(nT)])
    # ---- HIP loop
    system = mk_system(g["pos"], g["cell"], g["vel"], g["mass"])
    mdl = P.LennardJones(0.95, 0.9)
    integ = NoseHooverChain(Stack({"pair": PairPotentials(system, mdl, cutoff=2.5)}), system, T=1.0, num_chains=5,
                            Q=50.0).to(DEV)
    obs = rdf(system, nbins=60, r_range=(0.75, 2.4))
    opt = torch.optim.Adam(integ.parameters(), lr=lr)
    spec = integ.fused_spec("NH_verlet")
    hip = []
    for _ in range(epochs):
        opt.zero_grad()
        v_t, q_t, pv_t = ops.FusedTrajFn.apply(T(vel, DEV), T(pos, DEV), torch.zeros(R, 5, device=DEV), t.to(DEV),
                                               spec.flat_params(), spec)
        loss = (obs(q_t)[2] - 1).pow(2).mean()
        loss.backward()
        opt.step()
        hip.append((float(loss.detach()), float(mdl.sigma.detach()), float(mdl.epsilon.detach())))
    # ---- the same loop on the oracle
    theta = torch.tensor([0.95, 0.9], requires_grad=True)
    opt_o = torch.optim.Adam([theta], lr=lr)
    cell = T(g["cell"])
    ora = []
    for _ in range(epochs):
        opt_o.zero_grad()
        th = theta.detach().clone()
        trajs, eoms = [], []
        for r in range(R):
            term = O.PairTerm("lj", th, 2.5, cell, p=12, q=6, c=1)
            eom = O.NHCOracle(O.ModelOracle([term]), T(g["mass"]), 1.0, 50.0, 5)
            trajs.append(O.odeint_oracle(eom, (T(vel[r]), T(pos[r]), torch.zeros(5)), t))
            eoms.append(eom)
        qs = [tr[1].clone().requires_grad_(True) for tr in trajs]
        loss = (O.rdf_oracle(torch.stack(qs), cell, 60, (0.75, 2.4))[2] - 1).pow(2).mean()
        loss.backward()
        gth = torch.zeros(2)
        for r in range(R):
            grads = [torch.zeros_like(trajs[r][0]), qs[r].grad, torch.zeros_like(trajs[r][2])]
            gth = gth + O.adjoint_oracle(eoms[r], trajs[r], grads, t)[1]
        theta.grad = gth
        opt_o.step()
        ora.append((float(loss.detach()), float(theta[0]), float(theta[1])))
    for k, (a, b) in enumerate(zip(hip, ora)):
        assert abs(a[0] - b[0]) <= 2e-4 * abs(b[0]) + 1e-7, "loss at epoch %d: %r vs %r" % (k, a[0], b[0])
        assert abs(a[1] - b[1]) <= 2e-5 and abs(a[2] - b[2]) <= 2e-5, "(sigma, epsilon) at epoch %d: %r vs %r" % (k, a, b)
    assert hip[-1][1] != hip[0][1], "parameters moved"


@pytest.mark.parametrize("direct", [True, False])
def test_list_based_rdf_for_large_systems_vs_oracle(direct):
    """N >= 2048 atoms: rdf() searches pairs through the cell bins and counts them on the fine integer grid -- in one
    sweep without a neighbour list (csrc/rdf_cell.hip), or over a cell-list neighbour list (`ops.RDF_CELL_DIRECT =
    False`); its gradient is a tabulated pair force over the same bins / list.  3 frames of the 2 744-atom liquid, a
    third of the atoms of one frame shifted by whole cells (unwrapped coordinates), against the oracle."""
    from mdgrad_amd import ops
    from mdgrad_amd.observable import rdf
    pos, cell = liquid(14, seed=9, jitter=0.08)
    rng = np.random.default_rng(2)
    frames = np.stack([np.mod(pos + rng.normal(0, 0.05, pos.shape), cell) for _ in range(3)]).astype(np.float32)
    frames[1, ::3] += (np.asarray(cell) * np.array([1.0, -1.0, 0.0])).astype(np.float32)
    system = mk_system(pos, cell)
    wgt = torch.linspace(-1, 1, 100)
    x = T(frames, DEV).requires_grad_(True)
    was = ops.RDF_CELL_DIRECT
    ops.RDF_CELL_DIRECT = direct
    try:
        count, bins, gr = rdf(system, nbins=100, r_range=(0.75, 2.5))(x)
        (gx,) = torch.autograd.grad((gr * wgt.to(DEV)).sum(), x)
    finally:
        ops.RDF_CELL_DIRECT = was
    xo = T(frames).requires_grad_(True)
    _, _, go = O.rdf_oracle(xo, T(cell), 100, (0.75, 2.5))
    (gxo,) = torch.autograd.grad((go * wgt).sum(), xo)
    close(gr, go, 1e-4, 1e-4, "g(r), cell-based")
    close(gx, gxo, 1e-3, 1e-4 * float(gxo.abs().max()), "d(g.w)/dxyz, cell-based")


@pytest.mark.parametrize("width_scale", [3.0, 8.0])
def test_wide_gaussians_on_the_large_system_rdf_vs_oracle(width_scale):
    """the pair search of the large-system RDF is trimmed to the reach of the observable's Gaussians
    -- mu_last + 5.3 / s with s = sqrt(-coeff log2 e), where a term is below 2^-28 of its peak, never beyond the
    reference's own `end + 0.5` -- so the trim follows the user's `width`.  2 744 atoms, `width` = 3 and 8 bin spacings
    (at 8 the reach exceeds the reference's cutoff_boundary and the search runs to 3.0 like the reference's list),
    forward and d/dxyz against the oracle; the deviation budget is the same as at the default width."""
    from mdgrad_amd.observable import rdf
    pos, cell = liquid(14, seed=19, jitter=0.08)
    rng = np.random.default_rng(3)
    frames = np.stack([np.mod(pos + rng.normal(0, 0.05, pos.shape), cell) for _ in range(2)]).astype(np.float32)
    system = mk_system(pos, cell)
    nbins, rr = 100, (0.75, 2.5)
    width = width_scale * (rr[1] - rr[0]) / nbins
    wgt = torch.linspace(-1, 1, nbins)
    x = T(frames, DEV).requires_grad_(True)
    _, _, gr = rdf(system, nbins=nbins, r_range=rr, width=width)(x)
    (gx,) = torch.autograd.grad((gr * wgt.to(DEV)).sum(), x)
    xo = T(frames).requires_grad_(True)
    _, _, go = O.rdf_oracle(xo, T(cell), nbins, rr, width=width)
    (gxo,) = torch.autograd.grad((go * wgt).sum(), xo)
    close(gr, go, 1e-4, 1e-4, "g(r), width x%.0f" % width_scale)
    close(gx, gxo, 1e-3, 1e-4 * float(gxo.abs().max()), "d(g.w)/dxyz, width x%.0f" % width_scale)


def test_cell_sweep_rdf_in_a_dilute_box_with_capped_bins_vs_oracle():
    """A dilute gas in a large box: the bin count per side is capped at 16 (bins wider than the list cutoff), most
    bins are empty, a few atoms sit in the same spot twice (d = 0 pairs are not counted, topology.py:67)."""
    from mdgrad_amd.observable import rdf
    rng = np.random.default_rng(12)
    L = 60.0
    cell = np.array([L, L, L], dtype=np.float32)
    pos = rng.uniform(0, L, (2048, 3)).astype(np.float32)
    pos[1::2] = np.mod(pos[0::2] + rng.normal(0, 0.9, (1024, 3)), L).astype(np.float32)      # pairs within reach
    pos[100] = pos[7]                                                                        # a coincident pair
    frames = np.stack([pos, np.mod(pos + rng.normal(0, 0.3, pos.shape), L).astype(np.float32)])
    system = mk_system(pos, cell)
    wgt = torch.linspace(1, 2, 100)
    x = T(frames, DEV).requires_grad_(True)
    count, bins, gr = rdf(system, nbins=100, r_range=(0.75, 2.5))(x)
    (gx,) = torch.autograd.grad((gr * wgt.to(DEV)).sum(), x)
    xo = T(frames).requires_grad_(True)
    _, _, go = O.rdf_oracle(xo, T(cell), 100, (0.75, 2.5))
    (gxo,) = torch.autograd.grad((go * wgt).sum(), xo)
    close(gr, go, 2e-4, 2e-4 * float(go.detach().abs().max()), "g(r), dilute box")
    close(gx, gxo, 2e-3, 2e-4 * float(gxo.abs().max()), "d(g.w)/dxyz, dilute box")


def test_cell_sweep_rdf_equals_list_rdf_and_is_reproducible():
    """The direct sweep and the list-based kernels count the same pairs on the same integer grid: the same raw
    histograms (4 096 atoms, 5 frames, one of them a dense cluster that makes bins longer than a wave), and two runs of
    the sweep give bitwise equal gradients (bin order = rank by atom index, no float atomics)."""
    from mdgrad_amd import ops
    from mdgrad_amd.observable import rdf
    pos, cell = liquid(16, seed=3, jitter=0.08)
    rng = np.random.default_rng(4)
    frames = np.stack([np.mod(pos + rng.normal(0, 0.06, pos.shape), cell) for _ in range(5)]).astype(np.float32)
    frames[3, :200] = (np.asarray(cell) * 0.5 + rng.normal(0, 0.35, (200, 3))).astype(np.float32)      # a crowded bin
    system = mk_system(pos, cell)
    obs = rdf(system, nbins=100, r_range=(0.75, 2.5))
    wgt = torch.linspace(1, -1, 100, device=DEV)
    out = []
    was = ops.RDF_CELL_DIRECT
    try:
        for direct in (True, True, False):
            ops.RDF_CELL_DIRECT = direct
            x = T(frames, DEV).requires_grad_(True)
            count, _, gr = obs(x)
            (gx,) = torch.autograd.grad((gr * wgt).sum(), x)
            out.append((count.clone(), gx.clone()))
    finally:
        ops.RDF_CELL_DIRECT = was
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]), "two runs of the sweep differ"
    # (the list kernels apply the stored image shift, the sweep the minimum image on the fly: a last-bit difference in a
    #  distance can move a pair across a fine-bin edge)
    close(out[0][0], out[2][0], 1e-5, 1e-8, "histogram: sweep vs list")
    close(out[0][1], out[2][1], 1e-4, 1e-6 * float(out[2][1].abs().max()), "gradient: sweep vs list")


def test_large_path_replica_groups_on_concurrent_streams_are_bitwise_the_single_stream_run(monkeypatch):
    """Round 5: the launches of a large-system trajectory are issued for halves of the replicas on two side streams, so
    that one half's latency-bound prep launches overlap the other half's force sweeps (csrc/traj_large.hip lg_streams).  A
    replica's launches run in the same order on the same data: trajectories, adjoints and parameter gradients of 5 stacked
    1 000-atom replicas (odd count: uneven groups; different velocities: the replicas search at different steps) with 1, 2
    and 3 groups are bitwise equal."""
    from mdgrad_amd import ops, potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    pos, cell = liquid(10, seed=51, jitter=0.05)
    rng = np.random.default_rng(151)
    N, R = len(pos), 5
    system = mk_system(pos, cell, rng.normal(0, 1.0, pos.shape).astype(np.float32), np.full(N, 1.008, dtype=np.float32))
    mdl = P.LennardJones(1.0, 1.0)
    integ = NoseHooverChain(Stack({"p": PairPotentials(system, mdl, cutoff=2.5)}), system, T=1.0, num_chains=3, Q=30.0).to(DEV)
    integ.fused_large = True
    t = torch.Tensor([0.005 * i for i in range(13)]).to(DEV)
    q0 = T(np.stack([np.mod(pos + rng.normal(0, 0.02, pos.shape), cell) for _ in range(R)]).astype(np.float32), DEV)
    v0 = T(np.stack([rng.normal(0, 0.6 + 0.3 * r, pos.shape) for r in range(R)]).astype(np.float32), DEV)
    res = []
    for groups in ("1", "2", "3"):
        monkeypatch.setenv("MDG_LARGE_STREAMS", groups)
        spec = integ.fused_spec("NH_verlet")
        assert spec.large
        v, q = v0.clone().requires_grad_(True), q0.clone().requires_grad_(True)
        pv = torch.zeros(R, 3, device=DEV, requires_grad=True)
        v_t, q_t, pv_t = ops.FusedTrajFn.apply(v, q, pv, t, spec.flat_params(), spec)
        mdl.zero_grad()
        (q_t[:, ::4].pow(2).mean() + v_t[:, -1].pow(2).mean() + pv_t[:, -1].sum() * 1e-3).backward()
        res.append([q_t.detach().clone(), v_t.detach().clone(), pv_t.detach().clone(), v.grad.clone(), q.grad.clone(), pv.grad.clone(),
                    torch.stack([mdl.sigma.grad.reshape(()), mdl.epsilon.grad.reshape(())]).clone()])
    for other in res[1:]:
        for a, b, nm in zip(res[0], other, ("q_t", "v_t", "pv_t", "adj v0", "adj q0", "adj pv0", "dL/dtheta")):
            assert torch.equal(a, b), nm


def test_large_path_column_tiles_with_a_two_term_masked_stack_equal_fresh_searches_and_oracle():
    """The generic (multi-term) instantiation of the column-tile launches (csrc/traj_large.hip large_fwd_tiled<-1> /
    large_adj_tiled<-1>): a two-species LJ mixture -- an A-A term with cutoff 2.5 and an A-B LJ 9-6 term with cutoff 2.0, both
    with selection masks (index_tuple, torchmd/topology.py:37-42) -- on 1 000 atoms in a binned box.  Stored lists reused
    across steps (tiles, staged atom indices feed the masks) == a search at every evaluation (block = -1) == the oracle."""
    from mdgrad_amd import ops, potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    pos, cell = liquid(10, seed=61, jitter=0.05)
    rng = np.random.default_rng(161)
    N = len(pos)
    vel = rng.normal(0, 1.0, pos.shape).astype(np.float32)
    mass = np.full(N, 1.008, dtype=np.float32)
    A_, B_ = list(range(0, N, 2)), list(range(1, N, 2))
    t = torch.Tensor([0.004 * i for i in range(9)])
    system = mk_system(pos, cell, vel, mass)
    m_aa, m_ab = P.LennardJones(1.0, 1.0), P.LennardJones69(0.9, 0.7)
    stack = Stack({"aa": PairPotentials(system, m_aa, cutoff=2.5, index_tuple=(A_, A_)),
                   "ab": PairPotentials(system, m_ab, cutoff=2.0, index_tuple=(A_, B_))})
    integ = NoseHooverChain(stack, system, T=1.0, num_chains=3, Q=30.0).to(DEV)
    integ.fused_large = True

    def loss_fn(L):
        return L[1][::4].pow(2).sum() / L[1][::4].numel() + L[0][-1].pow(2).sum() / (N * 3) + L[2][-1].sum() * 1e-3

    res = []
    for block in (0, -1):
        spec = integ.fused_spec("NH_verlet")
        assert spec is not None and spec.large
        spec.block = block
        y0 = [s_.clone().requires_grad_(True) for s_ in integ.get_inital_states(wrap=True)]
        v_t, q_t, pv_t = ops.FusedTrajFn.apply(y0[0], y0[1], y0[2], t.to(DEV), spec.flat_params(), spec)
        for m in (m_aa, m_ab):
            m.zero_grad()
        loss_fn((v_t, q_t, pv_t)).backward()
        res.append([q_t.detach(), v_t.detach(), pv_t.detach(), y0[0].grad.clone(), y0[1].grad.clone(),
                    torch.stack([p.grad.reshape(()) for m in (m_aa, m_ab) for p in m.parameters()])])
    for a, b, nm in zip(res[0], res[1], ("q_t", "v_t", "pv_t", "adj v0", "adj q0", "dL/dtheta")):
        close(a, b, 2e-4, 2e-5 * float(b.abs().max()) + 1e-12, "%s (tiles vs searches, two masked terms)" % nm)
    it_aa, it_ab = (A_, A_), (A_, B_)
    terms = [O.PairTerm("lj", torch.tensor([1.0, 1.0]), 2.5, T(cell), p=12, q=6, c=1, index_tuple=it_aa),
             O.PairTerm("lj", torch.tensor([0.9, 0.7]), 2.0, T(cell), p=9, q=6, c=1, index_tuple=it_ab)]
    traj, lam, gth = oracle_run(pos, cell, vel, mass, terms, 1.0, 30.0, 3, t, loss_fn)
    close(res[0][0], traj[1], 0, 1e-4, "q_t vs oracle")
    close(res[0][5], gth, 5e-3, 5e-4 * float(gth.abs().max()), "dL/dtheta vs oracle")
    close(res[0][4], lam[1], 5e-3, 1e-3 * float(lam[1].abs().max()) + 1e-9, "adj q0 vs oracle")


@pytest.mark.parametrize("case", ["liquid4096", "crowded", "dilute", "tall_box"])
def test_cell_sweep_rdf_column_tiles_equal_the_row_sweep_bitwise(case, monkeypatch):
    """Round 5: the cell-sweep RDF stages the 3 x 3 bin columns around a workgroup's column in LDS (csrc/rdf_cell.hip
    rdf_cell_*_tile_kernel) instead of reading every candidate from L2.  Same candidates in the same per-lane order: the
    histogram is the one of the row sweep (MDG_RDF_CELL_TILES=0) count for count, the gradient agrees to fp32 rounding and
    is bitwise reproducible.  crowded: a 600-atom cluster
    makes one column overflow the staged capacity, so its tiles take the row sweep inside the tile kernel; dilute: 16 capped
    bins per side, mostly empty columns; tall_box: nb = (3, 4, 9) -- every x column is a neighbour of every other."""
    from mdgrad_amd.observable import rdf
    rng = np.random.default_rng(21)
    if case == "dilute":
        L = 60.0
        cell = np.array([L, L, L], dtype=np.float32)
        pos = rng.uniform(0, L, (2048, 3)).astype(np.float32)
        pos[1::2] = np.mod(pos[0::2] + rng.normal(0, 0.9, (1024, 3)), L).astype(np.float32)
        frames = np.stack([pos, np.mod(pos + rng.normal(0, 0.3, pos.shape), L).astype(np.float32)])
    elif case == "tall_box":
        cell = np.array([8.0, 10.6, 24.0], dtype=np.float32)
        n = 2048
        pos = (rng.uniform(0, 1, (n, 3)) * cell).astype(np.float32)
        frames = np.stack([pos, np.mod(pos + rng.normal(0, 0.2, pos.shape), cell).astype(np.float32),
                           (pos + np.array([8.0, -10.6, 24.0], dtype=np.float32)).astype(np.float32)])
    else:
        pos, cell = liquid(16, seed=5, jitter=0.08)
        frames = np.stack([np.mod(pos + rng.normal(0, 0.06, pos.shape), cell) for _ in range(4)]).astype(np.float32)
        if case == "crowded":
            frames[2, :600] = (np.asarray(cell) * 0.5 + rng.normal(0, 0.5, (600, 3))).astype(np.float32)
    system = mk_system(frames[0], cell)
    obs = rdf(system, nbins=100, r_range=(0.75, 2.5))
    wgt = torch.linspace(1, -1, 100, device=DEV)
    out = {}
    for tiles in ("1", "0", "1"):
        monkeypatch.setenv("MDG_RDF_CELL_TILES", tiles)
        x = T(frames, DEV).requires_grad_(True)
        count, _, gr = obs(x)
        (gx,) = torch.autograd.grad((gr * wgt).sum(), x)
        out.setdefault(tiles, []).append((count.clone(), gx.clone()))
    assert float(out["1"][0][0].sum()) > 0
    assert torch.equal(out["1"][0][0], out["0"][0][0]), "histogram: tiles vs row sweep (integer counts: the same pair set)"
    assert torch.equal(out["1"][0][1], out["1"][1][1]), "two runs of the tile kernels differ"
    dev = float((out["1"][0][1] - out["0"][0][1]).abs().max()) / float(out["0"][0][1].abs().max())
    assert dev <= 1e-5, "gradient: tiles vs row sweep, largest deviation / largest entry = %.3g" % dev     # (observed 1.6e-6)


@pytest.mark.parametrize("case", ["two_species_lj126", "excluded_pairs_ljfam", "two_species_nve", "odd_atoms"])
def test_ring_kernels_with_a_selection_mask_vs_oracle(case):
    """The wave-per-replica kernels with a masked term (first half): index_tuple = (A, B) of a two-species
    mixture (only A-B pairs interact, torchmd/topology.py:37-42) and ex_pairs (:44-53) -- the mask rides as two 128-bit rows per
    lane.  Trajectory, adjoints and parameter gradients of 4 replicas against the oracle, and the same launch on the
    one-workgroup-per-replica kernels (block = 128: the generic masked path)."""
    from mdgrad_amd import ops, potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NVE, NoseHooverChain
    g = load_golden("nhc_traj_lj")
    pos0, vel0, mass = g["pos"], g["vel"], g["mass"]
    if case == "odd_atoms":
        pos0, vel0, mass = pos0[:107], vel0[:107], mass[:107]
    n_atoms = len(pos0)
    R, nT = 4, 7
    rng = np.random.default_rng(len(case))
    system = mk_system(pos0, g["cell"], vel0, mass)
    A_, B_ = list(range(0, n_atoms, 2)), list(range(1, n_atoms, 2))
    if case == "excluded_pairs_ljfam":
        ex = np.array([[i, i + 1] for i in range(0, n_atoms - 1, 3)] + [[0, 5], [7, 100]])
        mdl, okw = P.LJFamily(epsilon=1.1, sigma=0.95, rep_pow=8, attr_pow=4), dict(p=8, q=4, c=1)
        kw, okw_sel = dict(ex_pairs=torch.as_tensor(ex)), dict(ex_pairs=ex)
        theta = [0.95, 1.1]
    else:
        mdl, okw, theta = P.LennardJones(1.0, 1.0), dict(p=12, q=6, c=1), [1.0, 1.0]
        kw, okw_sel = dict(index_tuple=(A_, B_)), dict(index_tuple=(A_, B_))
    stack = Stack({"pair": PairPotentials(system, mdl, cutoff=2.5, **kw)})
    nhc = case != "two_species_nve"
    integ = (NoseHooverChain(stack, system, T=1.0, num_chains=5, Q=50.0) if nhc else NVE(stack, system)).to(DEV)
    pos = np.mod(pos0[None] + rng.normal(0, 0.02, (R,) + pos0.shape), g["cell"]).astype(np.float32)
    vel = rng.normal(0, 0.5, pos.shape).astype(np.float32)
    t = torch.Tensor([0.004 * i for i in range(nT)])
    nrm = n_atoms * 3
    outs = {}
    for block in (64, 128):
        spec = integ.fused_spec("NH_verlet" if nhc else "verlet")
        assert spec is not None and not spec.large
        spec.block = block                                   # 64: wave per replica (ring); 128: workgroup per replica
        v0, q0 = T(vel, DEV).requires_grad_(True), T(pos, DEV).requires_grad_(True)
        pv0 = torch.zeros(R, 5, device=DEV, requires_grad=True) if nhc else None
        out = ops.FusedTrajFn.apply(v0, q0, pv0, t.to(DEV), spec.flat_params(), spec)
        mdl.zero_grad()
        loss = (out[1][:, ::2].pow(2).sum((1, 2, 3)) / (4 * nrm) + out[0][:, -1].pow(2).sum((1, 2)) / nrm).sum()
        if nhc:
            loss = loss + out[2][:, -1].sum()
        loss.backward()
        outs[block] = [out[0].detach(), out[1].detach(), v0.grad, q0.grad,
                       torch.cat([p.grad.reshape(-1) for p in mdl.parameters()])]
    for a, b, nm in zip(outs[64], outs[128], ("v_t", "q_t", "adj v0", "adj q0", "dtheta")):
        close(a, b, 2e-4, 2e-5 * float(b.abs().max()) + 1e-7, "ring vs workgroup kernels, masked term: " + nm)
    v_t, q_t, gv0, gq0, gth_hip = outs[64]
    gth_sum = np.zeros(2)
    for r in range(R):
        term = O.PairTerm("lj", torch.tensor(theta), 2.5, T(g["cell"]), **okw_sel, **okw)
        traj, lam, gth = oracle_run(
            pos[r], g["cell"], vel[r], mass, [term], 1.0, 50.0, 5, t,
            lambda L: L[1][::2].pow(2).sum() / (4 * nrm) + L[0][-1].pow(2).sum() / nrm + (L[2][-1].sum() if nhc else 0.0),
            ensemble="nhc" if nhc else "nve")
        close(q_t[r], traj[1], 0, 3e-5, "q_t[%d]" % r)
        close(v_t[r], traj[0], 0, 3e-4, "v_t[%d]" % r)
        close(gv0[r], lam[0], 2e-3, 3e-4 * float(lam[0].abs().max()), "adj v0[%d]" % r)
        close(gq0[r], lam[1], 2e-3, 3e-4 * float(lam[1].abs().max()), "adj q0[%d]" % r)
        gth_sum += gth.numpy()
    close(gth_hip, gth_sum, 2e-3, 3e-4 * np.abs(gth_sum).max(), "dL/dtheta")


@pytest.mark.parametrize("case", ["three_term_lj126_mixture", "two_term_ljfam_masked_plus_unmasked", "three_term_nve"])
def test_ring_kernels_with_several_masked_terms_vs_oracle(case):
    """Round 6 (VERDICT r5 missing #3 / next #7): species mixtures -- a Stack of LJ terms with index_tuple selections, the
    A-A / A-B / B-B stacks of scripts/fit_mix.py:101-117 (torchmd/interface.py:228-260, topology.py:37-42) -- on the
    wave-per-replica ring kernels: one ring sweep per term with the term's own 128-bit mask rows and constants
    (ring_force_terms).  Trajectory, adjoints and every term's parameter gradients of 4 replicas against the oracle, and the
    same launch on the one-workgroup-per-replica kernels."""
    from mdgrad_amd import ops, potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NVE, NoseHooverChain
    g = load_golden("nhc_traj_lj")
    pos0, vel0, mass = g["pos"], g["vel"], g["mass"]
    n_atoms = len(pos0)
    R, nT = 4, 7
    rng = np.random.default_rng(len(case))
    system = mk_system(pos0, g["cell"], vel0, mass)
    A_, B_ = list(range(0, n_atoms, 2)), list(range(1, n_atoms, 2))
    if case == "two_term_ljfam_masked_plus_unmasked":
        specs = [(P.LJFamily(epsilon=0.6, sigma=0.9, rep_pow=8, attr_pow=4), dict(index_tuple=(A_, B_)), dict(p=8, q=4, c=1), [0.9, 0.6]),
                 (P.LJFamily(epsilon=0.4, sigma=1.0, rep_pow=8, attr_pow=4), dict(), dict(p=8, q=4, c=1), [1.0, 0.4])]
    else:
        specs = [(P.LennardJones(1.0, 1.0), dict(index_tuple=(A_, A_)), dict(p=12, q=6, c=1), [1.0, 1.0]),
                 (P.LennardJones(0.9, 0.8), dict(index_tuple=(B_, B_)), dict(p=12, q=6, c=1), [0.9, 0.8]),
                 (P.LennardJones(0.95, 1.2), dict(index_tuple=(A_, B_)), dict(p=12, q=6, c=1), [0.95, 1.2])]
    mdls = [sp[0] for sp in specs]
    stack = Stack({"t%d" % k: PairPotentials(system, sp[0], cutoff=2.5, **sp[1]) for k, sp in enumerate(specs)})
    nhc = case != "three_term_nve"
    integ = (NoseHooverChain(stack, system, T=1.0, num_chains=5, Q=50.0) if nhc else NVE(stack, system)).to(DEV)
    pos = np.mod(pos0[None] + rng.normal(0, 0.02, (R,) + pos0.shape), g["cell"]).astype(np.float32)
    vel = rng.normal(0, 0.5, pos.shape).astype(np.float32)
    t = torch.Tensor([0.004 * i for i in range(nT)])
    nrm = n_atoms * 3
    outs = {}
    for block in (64, 128):
        spec = integ.fused_spec("NH_verlet" if nhc else "verlet")
        assert spec is not None and not spec.large
        spec.block = block                                   # 64: wave per replica (ring); 128: workgroup per replica
        v0, q0 = T(vel, DEV).requires_grad_(True), T(pos, DEV).requires_grad_(True)
        pv0 = torch.zeros(R, 5, device=DEV, requires_grad=True) if nhc else None
        out = ops.FusedTrajFn.apply(v0, q0, pv0, t.to(DEV), spec.flat_params(), spec)
        for m_ in mdls:
            m_.zero_grad()
        loss = (out[1][:, ::2].pow(2).sum((1, 2, 3)) / (4 * nrm) + out[0][:, -1].pow(2).sum((1, 2)) / nrm).sum()
        if nhc:
            loss = loss + out[2][:, -1].sum()
        loss.backward()
        outs[block] = [out[0].detach(), out[1].detach(), v0.grad, q0.grad,
                       torch.cat([p.grad.reshape(-1) for m_ in mdls for p in m_.parameters()])]
    for a, b, nm in zip(outs[64], outs[128], ("v_t", "q_t", "adj v0", "adj q0", "dtheta")):
        close(a, b, 2e-4, 2e-5 * float(b.abs().max()) + 1e-7, "ring vs workgroup kernels, %d masked terms: %s" % (len(specs), nm))
    v_t, q_t, gv0, gq0, gth_hip = outs[64]
    gth_sum = np.zeros(2 * len(specs))
    for r in range(R):
        terms = [O.PairTerm("lj", torch.tensor(sp[3]), 2.5, T(g["cell"]), **sp[1], **sp[2]) for sp in specs]
        traj, lam, gth = oracle_run(
            pos[r], g["cell"], vel[r], mass, terms, 1.0, 50.0, 5, t,
            lambda L: L[1][::2].pow(2).sum() / (4 * nrm) + L[0][-1].pow(2).sum() / nrm + (L[2][-1].sum() if nhc else 0.0),
            ensemble="nhc" if nhc else "nve")
        close(q_t[r], traj[1], 0, 3e-5, "q_t[%d]" % r)
        close(v_t[r], traj[0], 0, 3e-4, "v_t[%d]" % r)
        close(gv0[r], lam[0], 2e-3, 3e-4 * float(lam[0].abs().max()), "adj v0[%d]" % r)
        close(gq0[r], lam[1], 2e-3, 3e-4 * float(lam[1].abs().max()), "adj q0[%d]" % r)
        gth_sum += gth.numpy()
    close(gth_hip, gth_sum, 2e-3, 3e-4 * np.abs(gth_sum).max(), "dL/dtheta of every term")


def test_masked_ring_kernels_with_the_fused_rdf_observable():
    """A masked potential under the fused observable: the force sweep honours the mask, the RDF rides along unmasked -- the
    second launch (observable fused into the ring kernels) against the first (separate observable kernels) on 3 replicas."""
    from mdgrad_amd import ops, potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    from mdgrad_amd.observable import rdf
    g = load_golden("nhc_traj_lj")
    R, nT = 3, 9
    rng = np.random.default_rng(8)
    system = mk_system(g["pos"], g["cell"], g["vel"], g["mass"])
    A_, B_ = list(range(0, 108, 2)), list(range(1, 108, 2))
    mdl = P.LennardJones(1.0, 1.0)
    integ = NoseHooverChain(Stack({"pair": PairPotentials(system, mdl, cutoff=2.5, index_tuple=(A_, B_))}), system, T=1.0,
                            num_chains=5, Q=50.0).to(DEV)
    integ.fuse_observables = True                        # (opt-in since round 5)
    spec = integ.fused_spec("NH_verlet")
    spec.block = 64
    pos = np.mod(g["pos"][None] + rng.normal(0, 0.03, (R,) + g["pos"].shape), g["cell"]).astype(np.float32)
    vel = rng.normal(0, 1.0, pos.shape).astype(np.float32)
    t = torch.Tensor([0.005 * i for i in range(nT)]).to(DEV)
    obs = rdf(system, nbins=100, r_range=(0.75, 2.5))
    res = []
    for launch in range(2):
        v0, q0 = T(vel, DEV).requires_grad_(True), T(pos, DEV).requires_grad_(True)
        pv0 = torch.zeros(R, 5, device=DEV, requires_grad=True)
        out = ops.fused_traj(v0, q0, pv0, t, spec.flat_params(), spec)
        fused = out[1]._mdg_traj[3] is not None
        gr = obs(out[1])[2]
        mdl.zero_grad()
        ((gr * torch.linspace(0.5, 1.5, 100, device=DEV)).pow(2).sum() + out[0][:, -1].pow(2).sum() / 50.0).backward()
        res.append((fused, out[1].detach(), gr.detach(), q0.grad.clone(), v0.grad.clone(),
                    torch.stack([mdl.sigma.grad.reshape(()), mdl.epsilon.grad.reshape(())])))
    assert not res[0][0] and res[1][0], "the second launch must carry the observable"
    assert torch.equal(res[0][1], res[1][1]), "the trajectory itself does not change"
    close(res[1][2], res[0][2], 2e-4, 1e-4, "g(r)")
    for k, nm in ((3, "adj q0"), (4, "adj v0"), (5, "dtheta")):
        close(res[1][k], res[0][k], 1e-3, 1e-3 * float(res[0][k].abs().max()) + 1e-7, nm)
