"""f4: BondPotentials / AnglePotentials as HIP kernels (csrc/bonded.hip, mdg_bonded_eval) behind the
reference's classes (torchmd/interface.py:406-510) -- energy, force and the Hessian-vector product against the reference's
golden vectors (G12, G18), and the polymer Stack of demo/fold.py:131-161 on the analytic adjoint + HIP-graph replay."""
import numpy as np
import pytest
import torch

import oracle as O
from conftest import load_golden
from test_gpu_parity import T, close, mk_system, DEV

pytestmark = pytest.mark.gpu


def _terms(g, system):
    from mdgrad_amd.interface import AnglePotentials, BondPotentials
    return {"bond": BondPotentials(system, torch.as_tensor(g["bonds"]), float(g["k_bond"]), float(g["ro"])),
            "angle": AnglePotentials(system, torch.as_tensor(g["angles"]), float(g["k_angle"]), float(g["theta0"]))}


@pytest.mark.parametrize("tag", ["bond", "angle"])
def test_bonded_kernel_energy_force_hvp_golden(tag):
    """One launch of mdg_bonded_eval: U, F = -dU/dx and d(w.F)/dx = -H w against the reference (autograd, double autograd);
    the same through autograd (BondedEnergyFn -> BondedGradFn: the reference's create_graph path), and added onto the
    buffers of another Stack member."""
    from mdgrad_amd import ops
    g, h = load_golden("bonded"), load_golden("bonded_hvp")
    system = mk_system(g["pos"], g["cell"])
    mod = _terms(g, system)[tag]
    assert mod.supports_force_vjp() and mod.supports_static_topology()
    q, w = T(g["pos"], DEV), T(h["w"], DEV)
    # (the chain of G12 has every bond at |b|^2 = ro to rounding: its bond forces are the fp32 noise of 2 k (|b|^2 - ro) b, so
    #  the force scale is that of the factors, 2 k ro |b|; the trajectories of G18 stretch the bonds)
    fmax = max(float(np.abs(g[tag + "_force"]).max()), 2.0 * float(g["k_bond"]) * float(g["ro"]) * 1.1 if tag == "bond" else 0.0)
    hmax = float(np.abs(h[tag + "_hw"]).max())
    close(mod(q).reshape(1), g[tag + "_energy"], 1e-5, 1e-5, tag + " energy")
    close(mod.force(q), g[tag + "_force"], 1e-4, 1e-5 * fmax, tag + " force")
    F, dq, gth = mod.force_vjp(q, w)
    assert gth == []
    close(F, g[tag + "_force"], 1e-4, 1e-5 * fmax, tag + " force (vjp launch)")
    close(-dq, h[tag + "_hw"], 1e-4, 2e-5 * hmax, tag + " H.w")
    # autograd route: first and second order
    x = q.clone().requires_grad_(True)
    (gq,) = torch.autograd.grad(mod(x), x, create_graph=True)
    close(-gq, g[tag + "_force"], 1e-4, 1e-5 * fmax, tag + " force (autograd)")
    (hw,) = torch.autograd.grad((gq * w).sum(), x)
    close(hw, h[tag + "_hw"], 1e-4, 2e-5 * hmax, tag + " H.w (double autograd)")
    # accumulation onto existing buffers (Stack.force / force_vjp hand the running sums down)
    F0, D0 = torch.randn_like(q), torch.randn_like(q)
    F1, D1, _ = mod.force_vjp(q, w, into=(F0.clone(), D0.clone()))
    close(F1 - F0, g[tag + "_force"], 1e-4, 1e-5 * fmax + 1e-6, tag + " force added onto a buffer")
    close(-(D1 - D0), h[tag + "_hw"], 1e-4, 2e-5 * hmax + 1e-6, tag + " H.w added onto a buffer")
    # the oracle restatement on the same inputs (bit-level agreement is not expected: fp32 summation order)
    term = (O.BondTerm(g["bonds"], float(g["k_bond"]), float(g["ro"]), T(g["cell"])) if tag == "bond"
            else O.AngleTerm(g["angles"], float(g["k_angle"]), float(g["theta0"]), T(g["cell"])))
    Fo, dqo, _ = term.force_vjp(T(g["pos"]), T(h["w"]))
    close(F, Fo, 1e-4, 1e-5 * fmax, tag + " force vs oracle")
    close(dq, dqo, 1e-4, 2e-5 * hmax, tag + " d(w.F)/dx vs oracle")
    two = ops.bonded_eval(mod.table(), q, w=w)
    assert torch.equal(two["grad"], -F) and torch.equal(two["hw"], -dq), "two launches are bitwise equal"


def test_bonded_image_flags_follow_get_offsets():
    """topology.get_offsets (topology.py:75-80) is NON-strict on the upper side: a bond component of exactly +L/2 is
    folded, one of exactly -L/2 is not -- unlike the neighbour list's strict test."""
    from mdgrad_amd.interface import BondPotentials
    L = 4.0
    pos = np.array([[3.0, 1.0, 1.0], [1.0, 1.0, 1.0], [0.5, 3.0, 1.0], [2.5, 3.0, 1.0]], dtype=np.float32)
    system = mk_system(pos, np.array([L, L, L], dtype=np.float32))
    top = torch.LongTensor([[0, 1], [2, 3]])            # b_x = +2 = L/2 (folded to -2) and b_x = -2 (kept)
    mod = BondPotentials(system, top, 1.5, 1.0)
    term = O.BondTerm(top.numpy(), 1.5, 1.0, torch.tensor([L, L, L]))
    q = T(pos, DEV)
    close(mod(q).reshape(()), term.energy(T(pos)), 1e-6, 1e-6, "energy")
    close(mod.force(q), term.force(T(pos)), 1e-5, 1e-6, "force")
    assert float(mod.force(q)[0, 0]) > 0 and float(mod.force(q)[2, 0]) > 0, "both bonds point along -x after folding"


def _polymer(g, tag):
    from mdgrad_amd import potentials as P
    from mdgrad_amd.interface import BondPotentials, GNNPotentials, PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    from mdgrad_amd.nn import get_model
    from test_gpu_schnet import params_of, sd_of
    system = mk_system(g["pos"], g["cell"], g["vel"], g["masses"], g["numbers"])
    bonds = torch.as_tensor(g["bonds"])
    bond = BondPotentials(system, bonds, float(g["k_bond"]), float(g["ro"]))
    pair = PairPotentials(system, P.ExcludedVolume(float(g["pair_sigma"]), float(g["pair_epsilon"]), int(g["pair_power"])),
                          cutoff=float(g["pair_cutoff"]), ex_pairs=bonds)                  # demo/fold.py:150-155
    if tag == "fold":
        net = get_model(params_of(g))
        net.load_state_dict(sd_of(g))
        stack = Stack({"gnn": GNNPotentials(system, net, cutoff=float(g["cutoff"])), "prior": bond, "pair": pair})
    else:
        stack = Stack({"pair": pair, "bond": bond})
    integ = NoseHooverChain(stack, system, T=float(g["T"]), num_chains=int(g["chains"]), Q=float(g["Q"]), adjoint=True).to(DEV)
    return system, integ


@pytest.mark.parametrize("tag", ["pairbond", "fold"])
@pytest.mark.parametrize("graphs_on", [True, False], ids=["graph_replay", "eager"])
def test_polymer_stack_trajectory_adjoint_golden(tag, graphs_on):
    """demo/fold.py:131-161: Stack{gnn, prior = BondPotentials, pair = ExcludedVolume(power 10) without the bonded pairs},
    and Stack{pair, bond}: NoseHooverChain trajectory and adjoint against the reference's (golden G18).  The stack takes
    the analytic adjoint (no autograd double backward) and, by default, HIP-graph replay."""
    from mdgrad_amd import graphs
    from mdgrad_amd.sovlers import odeint_adjoint
    g = load_golden("fold_traj")
    system, integ = _polymer(g, tag)
    assert integ.fused_spec("NH_verlet") is None
    assert integ.model.supports_force_vjp() and integ.supports_rhs_vjp(), "a polymer Stack must not fall to the autograd branch"
    assert graphs.enabled(integ)
    integ.use_graphs = graphs_on
    assert [n for n, _ in integ.named_parameters()] == [str(x) for x in g[tag + "_param_names"]]
    y0 = [s.clone().requires_grad_(True) for s in integ.get_inital_states(wrap=True)]
    nT = g[tag + "_q_t"].shape[0]
    t = torch.Tensor([float(g["dt"]) * i for i in range(nT)]).to(DEV)
    calls = {"n": 0}
    orig = integ.model.force_vjp

    def counted(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    integ.model.force_vjp = counted
    v_t, q_t, pv_t = odeint_adjoint(integ, tuple(y0), t, method="NH_verlet")
    for x, k in zip((v_t, q_t, pv_t), ["v_t", "q_t", "pv_t"]):
        close(x, g[tag + "_" + k], 1e-4, 1e-4 * max(1e-3, np.abs(g[tag + "_" + k]).max()), tag + " " + k)
    loss = q_t[::3].pow(2).mean() * 1e-2 + v_t[-1].pow(2).mean() + pv_t[-1].sum() * 1e-2
    close(loss.reshape(1), g[tag + "_loss"], 1e-4, 1e-6, tag + " loss")
    loss.backward()
    assert calls["n"] > 0, "the adjoint did not go through force_vjp"
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in integ.parameters()])
    ref = g[tag + "_grad_flat"]
    close(flat, ref, 5e-3, 2e-4 * np.abs(ref).max(), tag + " dL/dtheta")
    for x, k in zip(y0, ["grad_v0", "grad_q0", "grad_pv0"]):
        close(x.grad, g[tag + "_" + k], 5e-3, 2e-3 * np.abs(g[tag + "_" + k]).max(), tag + " " + k)


def test_angle_term_in_a_stack_trajectory_vs_oracle():
    """AnglePotentials has no _reset_topology in the reference (it cannot be a Stack member there); here it can: Stack(pair +
    bond + angle), 10 NHC steps + adjoint against the oracle (whose angle term is pinned to the reference's energy / force /
    H.w by tests/test_oracle_golden.py)."""
    from mdgrad_amd import potentials as P
    from mdgrad_amd.interface import AnglePotentials, BondPotentials, PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    from mdgrad_amd.sovlers import odeint_adjoint
    from test_gpu_parity import oracle_run
    g = load_golden("fold_traj")
    n = g["pos"].shape[0]
    angles = np.array([[i, i + 1, i + 2] for i in range(n - 2)])
    system = mk_system(g["pos"], g["cell"], g["vel"], g["masses"], g["numbers"])
    bonds = torch.as_tensor(g["bonds"])
    mdl = P.ExcludedVolume(0.9, 0.5, 10)
    stack = Stack({"pair": PairPotentials(system, mdl, cutoff=2.5, ex_pairs=bonds),
                   "bond": BondPotentials(system, bonds, 3.0, 1.21),
                   "angle": AnglePotentials(system, torch.as_tensor(angles), 2.0, 1.9)})
    integ = NoseHooverChain(stack, system, T=0.5, num_chains=5, Q=50.0, adjoint=True).to(DEV)
    assert integ.supports_rhs_vjp()
    t = torch.Tensor([0.005 * i for i in range(11)])
    y0 = [s.clone().requires_grad_(True) for s in integ.get_inital_states(wrap=True)]
    v_t, q_t, pv_t = odeint_adjoint(integ, tuple(y0), t.to(DEV), method="NH_verlet")

    def loss_fn(L):
        return L[1][::2].pow(2).mean() * 1e-2 + L[0][-1].pow(2).mean() + L[2][-1].sum() * 1e-2
    loss_fn((v_t, q_t, pv_t)).backward()
    cell = T(g["cell"])
    terms = [O.PairTerm("lj", torch.tensor([0.9, 0.5]), 2.5, cell, ex_pairs=g["bonds"], p=10, q=0, c=0),
             O.BondTerm(g["bonds"], 3.0, 1.21, cell), O.AngleTerm(angles, 2.0, 1.9, cell)]
    traj, lam, gth = oracle_run(g["pos"], g["cell"], g["vel"], g["masses"], terms, 0.5, 50.0, 5, t, loss_fn)
    close(q_t, traj[1], 0, 2e-5, "q_t")
    close(v_t, traj[0], 1e-3, 1e-4 * float(traj[0].abs().max()), "v_t")
    close(pv_t, traj[2], 2e-3, 1e-5, "pv_t")
    for x, l, nm in zip(y0, lam, ("adj v0", "adj q0", "adj pv0")):
        close(x.grad, l, 5e-3, 2e-3 * float(l.abs().max()) + 1e-9, nm)
    got = torch.stack([mdl.sigma.grad.reshape(()), mdl.epsilon.grad.reshape(())])
    close(got, gth, 5e-3, 5e-4 * float(gth.abs().max()), "dL/d(sigma, epsilon)")
