"""Size-independent properties at the launch sizes BASELINE.json's metric is quoted on (the oracle cannot follow there: a
16 384-replica launch is 800 k step-replicas):

  * replica permutation: replicas never interact, so a launch with the replicas permuted must return the permuted results
    BIT FOR BIT -- whatever wave, workgroup, XCD or replica group a replica lands on (a launch-geometry bug at high
    workgroup indices, a race between replica groups, a read past a replica's rows all break it);
  * linearity of the adjoint (torchmd/sovlers.py:211-293 is linear in the incoming frame gradients):
    adj(g1 + g2) = adj(g1) + adj(g2) to fp32 rounding;
  * and one replica of the full launch -- the last -- against its own oracle run over the whole horizon.

Headline geometry (configs #1/#2: 108-atom LJ, 16 384 replicas x 49 steps) and config #4's (64 x 4 096 atoms)."""
import numpy as np
import pytest
import torch

import oracle as O
from conftest import load_golden
from test_gpu_parity import T, close, lj_setup, liquid, oracle_run, DEV
from test_gpu_secondary_pins import _lj_large

pytestmark = pytest.mark.gpu


def _launch(spec, vel, pos, chains, t, gq=None, gv=None):
    """One forward launch + (with frame gradients) one adjoint launch; detached results."""
    from mdgrad_amd import ops
    R = pos.shape[0]
    v0, q0 = vel.clone().requires_grad_(True), pos.clone().requires_grad_(True)
    pv0 = torch.zeros(R, chains, device=DEV, requires_grad=True)
    v_t, q_t, pv_t = ops.FusedTrajFn.apply(v0, q0, pv0, t, spec.flat_params(), spec)
    out = [v_t.detach(), q_t.detach(), pv_t.detach()]
    if gq is not None:
        for p_ in spec._integrator.parameters():
            p_.grad = None
        torch.autograd.backward([q_t, v_t], [gq, gv])
        out += [v0.grad, q0.grad, pv0.grad, torch.cat([p_.grad.reshape(-1) for p_ in spec._integrator.parameters()])]
    return out


def test_headline_launch_16384_replicas_permutation_linearity_and_last_replica_vs_oracle():
    g = load_golden("nhc_traj_lj")
    system, mdl, integ = lj_setup(g)
    spec = integ.fused_spec("NH_verlet")
    assert spec is not None and not spec.large
    R, nT, N = 16384, 50, 108
    gen = torch.Generator(device=DEV).manual_seed(7)
    rn = lambda *s: torch.randn(*s, generator=gen, device=DEV)
    base = torch.from_numpy(np.asarray(g["pos"], dtype=np.float32)).to(DEV)
    cell = torch.from_numpy(np.asarray(g["cell"], dtype=np.float32)).to(DEV)
    pos = torch.remainder(base[None] + 0.03 * rn(R, N, 3), cell)
    vel = rn(R, N, 3)
    t = torch.Tensor([0.005 * i for i in range(nT)]).to(DEV)
    g1q = rn(R, nT, N, 3) * 1e-3
    g1v = torch.zeros(R, nT, N, 3, device=DEV)
    g1v[:, -1] = 1e-3 * rn(R, N, 3)
    a = _launch(spec, vel, pos, 5, t, g1q, g1v)
    # --- the same launch again: bitwise (no atomics, fixed orders)
    b = _launch(spec, vel, pos, 5, t, g1q, g1v)
    for x, y, nm in zip(a, b, ("v_t", "q_t", "pv_t", "adj v0", "adj q0", "adj pv0", "dtheta")):
        assert torch.equal(x, y), "two identical 16 384-replica launches differ in " + nm
    del b
    # --- replicas permuted
    perm = torch.randperm(R, generator=gen, device=DEV)
    c = _launch(spec, vel[perm], pos[perm], 5, t, g1q[perm], g1v[perm])
    for x, y, nm in zip(a[:6], c[:6], ("v_t", "q_t", "pv_t", "adj v0", "adj q0", "adj pv0")):
        assert torch.equal(x[perm], y), "permuting the replicas of the launch changes " + nm
    close(c[6], a[6], 1e-5, 1e-6 * float(a[6].abs().max()), "dtheta (a sum over the replicas: order-dependent rounding only)")
    del c
    # --- linearity of the adjoint in the frame gradients
    g2q = rn(R, nT, N, 3) * 1e-3
    g2v = torch.zeros_like(g1v)
    a2 = _launch(spec, vel, pos, 5, t, g2q, g2v)
    a12 = _launch(spec, vel, pos, 5, t, g1q + g2q, g1v + g2v)
    for k, nm in ((3, "adj v0"), (4, "adj q0"), (5, "adj pv0"), (6, "dtheta")):
        s = a[k] + a2[k]
        close(a12[k], s, 2e-4, 2e-5 * float(s.abs().max()), "adjoint linearity, " + nm)
    # --- the LAST replica of the launch over the whole 49 steps vs its own oracle run
    r = R - 1
    term = O.PairTerm("lj", torch.tensor([1.0, 1.0]), float(g["cutoff"]), T(g["cell"]), p=12, q=6, c=1)
    tt = t.cpu()

    def loss_one(L):
        return (L[1] * g1q[r].cpu()).sum() + (L[0] * g1v[r].cpu()).sum()

    traj, lam, _ = oracle_run(pos[r].cpu().numpy(), g["cell"], vel[r].cpu().numpy(), g["mass"], [term], float(g["T"]), float(g["Q"]),
                              int(g["chains"]), tt, loss_one)
    close(a[1][r], traj[1], 0, 1e-4, "q_t of replica 16 383 (49 steps)")
    close(a[0][r], traj[0], 0, 2e-3, "v_t of replica 16 383")
    for k, l, nm in ((3, lam[0], "adj v0"), (4, lam[1], "adj q0")):
        close(a[k][r], l, 5e-3, 2e-3 * float(l.abs().max()), nm + " of replica 16 383")


def test_config4_launch_64_x_4096_atoms_permutation_and_linearity():
    base, cell = liquid(16, seed=44, jitter=0.05)
    N, R, nT = len(base), 64, 7
    mdl, integ, mass = _lj_large(base, cell)
    spec = integ.fused_spec("NH_verlet")
    assert spec.large
    gen = torch.Generator(device=DEV).manual_seed(11)
    rn = lambda *s: torch.randn(*s, generator=gen, device=DEV)
    cl = torch.from_numpy(cell).to(DEV)
    pos = torch.remainder(torch.from_numpy(base).to(DEV)[None] + 0.02 * rn(R, N, 3), cl)
    scale = torch.tensor([0.8 + 0.4 * (r % 3) for r in range(R)], device=DEV)[:, None, None]
    vel = rn(R, N, 3) * scale
    t = torch.Tensor([0.005 * i for i in range(nT)]).to(DEV)
    g1q = rn(R, nT, N, 3) * 1e-3
    g1v = torch.zeros(R, nT, N, 3, device=DEV)
    g1v[:, -1] = 1e-3 * rn(R, N, 3)
    a = _launch(spec, vel, pos, 3, t, g1q, g1v)
    perm = torch.randperm(R, generator=gen, device=DEV)      # (replicas change their replica group and stream, too)
    c = _launch(spec, vel[perm], pos[perm], 3, t, g1q[perm], g1v[perm])
    for x, y, nm in zip(a[:6], c[:6], ("v_t", "q_t", "pv_t", "adj v0", "adj q0", "adj pv0")):
        assert torch.equal(x[perm], y), "permuting the replicas of the 64 x 4 096-atom launch changes " + nm
    close(c[6], a[6], 1e-5, 1e-6 * float(a[6].abs().max()), "dtheta")
    del c
    g2q = rn(R, nT, N, 3) * 1e-3
    a2 = _launch(spec, vel, pos, 3, t, g2q, torch.zeros_like(g1v))
    a12 = _launch(spec, vel, pos, 3, t, g1q + g2q, g1v)
    for k, nm in ((3, "adj v0"), (4, "adj q0"), (5, "adj pv0"), (6, "dtheta")):
        s = a[k] + a2[k]
        close(a12[k], s, 2e-4, 2e-5 * float(s.abs().max()), "adjoint linearity, " + nm)


def test_config5_stack_8_x_4096_beads_replica_permutation():
    """The stack bench.py's schnet4096 leg times (8 x 4 096 beads, bf16 filter operands + bf16 gathered node rows, built by the
    same function): 4 steps + per-replica RDF loss + adjoint with the replicas of the stack permuted -- every replica's
    trajectory and the adjoint of its initial state are those of the unpermuted stack (the interaction-block kernels work atom
    by atom and edge by edge: which workgroup, XCD window or 16-edge tile a replica's rows land in must not matter); the
    parameter gradient, a sum over all atoms, to rounding."""
    import bench
    from mdgrad_amd import units
    from test_gpu_secondary_pins import _run_schnet_workload
    R = 8
    wl = bench.build_schnet_workload(DEV, R, True, 2000, rows16=True)
    N = wl["N"]
    pos = wl["system"].get_positions().reshape(R, N, 3).copy()
    vel = wl["system"].get_velocities().reshape(R, N, 3).copy()
    t = torch.Tensor([units.fs * i for i in range(5)])
    a = _run_schnet_workload(wl, t, 2)
    perm = np.array([5, 2, 7, 0, 3, 6, 1, 4])
    wl["system"].set_positions(pos[perm].reshape(-1, 3))
    wl["system"].set_velocities(vel[perm].reshape(-1, 3))
    b = _run_schnet_workload(wl, t, 2)
    pd = torch.from_numpy(perm).to(DEV)
    for k, x, y in (("q_t", a["q"][:, pd], b["q"]), ("v_t", a["v"][:, pd], b["v"]), ("pv_t", a["pv"][:, pd], b["pv"]),
                    ("adj q0", a["gq0"][pd], b["gq0"]), ("adj v0", a["gv0"][pd], b["gv0"])):
        assert torch.equal(x, y), "permuting the replicas of the stack changes " + k      # (bit for bit on MI355X)
    close(b["flat"], a["flat"], 1e-4, 1e-5 * float(a["flat"].abs().max()), "dL/dtheta")
