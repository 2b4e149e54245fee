"""Golden-vector generator (BUILD CONTAINER ONLY).

Imports the reference at /root/reference (with the ase/xitorch stand-ins of
_ase_stub.py), runs its CPU PyTorch path on explicit, seeded inputs and writes
small .npz fixtures next to this file.  Only DATA (inputs + reference outputs)
is committed; no reference source travels.

    python tests/golden/make_goldens.py

Golden sets:
  G1 nbr_*      generate_nbr_list            torchmd/topology.py:30-73
  G2 pair_*     compute_dis + pair forms     torchmd/topology.py:5-12, potentials.py
  G3 nhc_rhs    NoseHooverChain.forward      torchmd/md.py:210-240
  G4 nhc_traj   odeint(NH_verlet)            torchmd/sovlers.py:106-127,171-193
  G5 nhc_adj    odeint_adjoint + rdf + bwd   torchmd/sovlers.py:196-293
  G6 nve_*      NVE / verlet fwd + adjoint   torchmd/md.py:98-157, sovlers.py:21-104
  G7 rdf_*      rdf observable               torchmd/observable.py:33-76
  G8 schnet_*   SchNet energy/forces/hvp     nff/nn/models/schnet.py:23-171
  G9 gnn_traj   Stack(GNN+pair) NHC + adj    torchmd/interface.py:86-136,364-403
  G10 sim_*     Simulations 2 epochs         torchmd/md.py:14-96
  G13 exp_rdf   get_exp_rdf target normalisation                   scripts/data.py:11-31
  G12 bonded    BondPotentials / AnglePotentials energy + forces   torchmd/interface.py:406-510
  G14 gnn_traj_water192  config #3: 192-atom water, SchNet A128/F128/G32/3 conv + prior, NHC + adjoint
  G15 schnet_cg64_wide   SchNet A64/F128/G30/2 conv (config #5 widths): U, F, H.w, d(w.F)/dtheta
  G16 vacf_temp          vacf / Temperature observables      torchmd/observable.py:153-163, thermo.py:57-66
  G17 schnet_cg64_a256   SchNet A256/F256/G41/2 conv (wide search-space setting): U, F, H.w, d(w.F)/dtheta
  G18 bonded_hvp, fold_traj   f4 in the adjoint: H.w of the bonded terms; the polymer Stack of demo/fold.py:131-161
                                             (GNN + BondPotentials + ExcludedVolume with the bonded pairs excluded) and
                                             Stack(pair + bond): NHC trajectory + adjoint      torchmd/interface.py:406-510
  G11 pair_mlp  pairMLP / TpairMLP energies, forces, Stack(pairMLP + LJFamily) NHC trajectory + adjoint
                                             torchmd/potentials.py:163-217, interface.py:139-215
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ase_stub  # noqa: E402

_ase_stub.install()
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402
from torchmd.topology import generate_nbr_list, compute_dis  # noqa: E402
from torchmd.system import System  # noqa: E402
from torchmd import potentials as P  # noqa: E402
from torchmd.interface import (PairPotentials, TPairPotentials, GNNPotentials, Stack,  # noqa: E402
                               BondPotentials, AnglePotentials)
from torchmd.md import NoseHooverChain, NVE, Simulations  # noqa: E402
from torchmd.sovlers import odeint, odeint_adjoint  # noqa: E402
from torchmd.observable import rdf  # noqa: E402
from nff.nn.models.schnet import SchNet  # noqa: E402

torch.set_num_threads(4)
F32 = np.float32


def fcc(size, a):
    basis = np.array([[0, 0, 0], [.5, .5, 0], [.5, 0, .5], [0, .5, .5]])
    pts = []
    for i in range(size):
        for j in range(size):
            for k in range(size):
                for b in basis:
                    pts.append((np.array([i, j, k]) + b) * a)
    return np.array(pts), np.array([a * size] * 3)


def make_system(pos, cell, numbers=None, masses=None, vel=None):
    n = len(pos)
    masses = np.full(n, 1.008) if masses is None else masses
    atoms = _ase_stub.Atoms(positions=pos, cell=cell, numbers=numbers, masses=masses)
    system = System(atoms, device="cpu")
    if vel is not None:
        system.set_velocities(vel)
    return system


def lj_inputs(seed=0, jitter=0.05, size=3, a=1.6, T=1.0):
    rng = np.random.default_rng(seed)
    pos, cell = fcc(size, a)
    pos = pos + rng.uniform(-jitter, jitter, pos.shape)
    pos = np.mod(pos, cell)                       # keep strictly in the cell
    pos = pos.astype(F32).astype(np.float64)      # fp32-representable inputs
    vel = rng.normal(0, np.sqrt(T / 1.008), pos.shape).astype(F32).astype(np.float64)
    return pos, cell, vel


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


# ------------------------------------------------------------------ G1
def g1():
    pos, cell, _ = lj_inputs(seed=1)
    xyz = torch.Tensor(pos)
    nbr, dis, off = generate_nbr_list(xyz, 2.5, torch.Tensor(cell), get_dis=True)
    save("nbr_diag108", xyz=xyz, cell=cell.astype(F32), cutoff=2.5, nbr=nbr, offsets=off, dis=dis)

    # exact lattice: separations of exactly L/2 exercise the strict >0.5 test
    lat, cell = fcc(3, 1.6)
    xyz = torch.Tensor(lat)
    nbr, dis, off = generate_nbr_list(xyz, 2.5, torch.Tensor(cell), get_dis=True)
    save("nbr_lattice108", xyz=xyz, cell=cell.astype(F32), cutoff=2.5, nbr=nbr, offsets=off, dis=dis)

    # triclinic 3x3 cell
    rng = np.random.default_rng(2)
    cellm = np.array([[6.0, 0.0, 0.0], [1.2, 5.5, 0.0], [0.7, -0.9, 6.3]])
    frac = rng.uniform(0, 1, (64, 3))
    xyz = torch.Tensor(frac @ cellm)
    cellt = torch.Tensor(cellm)
    nbr, dis, off = generate_nbr_list(xyz, 2.2, cellt, get_dis=True)
    save("nbr_tric64", xyz=xyz, cell=cellt, cutoff=2.2, nbr=nbr, offsets=off, dis=dis)

    # species selection + exclusions
    pos, cell, _ = lj_inputs(seed=3)
    xyz = torch.Tensor(pos)
    idx_a = list(range(0, 108, 2))
    idx_b = list(range(1, 108, 2))
    ex = torch.LongTensor([[0, 1], [2, 3], [10, 50], [4, 7]])
    nbr, dis, off = generate_nbr_list(xyz, 2.5, torch.Tensor(cell), index_tuple=(idx_a, idx_b),
                                      ex_pairs=ex, get_dis=True)
    save("nbr_mask108", xyz=xyz, cell=cell.astype(F32), cutoff=2.5, idx_a=idx_a, idx_b=idx_b,
         ex_pairs=ex, nbr=nbr, offsets=off, dis=dis)
    nbr, dis, off = generate_nbr_list(xyz, 2.5, torch.Tensor(cell), index_tuple=(idx_a, idx_a),
                                      get_dis=True)
    save("nbr_self108", xyz=xyz, cell=cell.astype(F32), cutoff=2.5, idx_a=idx_a, idx_b=idx_a,
         nbr=nbr, offsets=off, dis=dis)

    # batched frames (rdf usage)
    rng = np.random.default_rng(4)
    frames = np.stack([np.mod(pos + rng.normal(0, 0.1, pos.shape), cell) for _ in range(3)])
    xyz = torch.Tensor(frames)
    nbr, dis, off = generate_nbr_list(xyz, 3.0, torch.Tensor(cell), get_dis=True)
    # (the reference's batched `offsets` gather is mis-indexed and unused by rdf: not recorded)
    save("nbr_batched", xyz=xyz, cell=cell.astype(F32), cutoff=3.0, nbr=nbr, dis=dis)


# ------------------------------------------------------------------ G2
PAIR_FORMS = {
    "lj": lambda: P.LennardJones(sigma=1.05, epsilon=0.9),
    "ljfam_8_4": lambda: P.LJFamily(sigma=0.95, epsilon=1.1, attr_pow=4, rep_pow=8),
    "lj69": lambda: P.LennardJones69(sigma=1.0, epsilon=1.2),
    "exvol12": lambda: P.ExcludedVolume(sigma=1.0, epsilon=1.0, power=12),
    "exvol10": lambda: P.ExcludedVolume(sigma=1.1, epsilon=0.7, power=10),
    "morse_pos": lambda: P.ModifiedMorse(a=3.0, phi=1.5),
    "morse_neg": lambda: P.ModifiedMorse(a=2.5, phi=-1.2),
    "buck": lambda: P.Buck(A=1000.0, B=3.5, C=5.0),
}


def g2():
    pos, cell, _ = lj_inputs(seed=5)
    system = make_system(pos, cell)
    rng = np.random.default_rng(6)
    w = torch.Tensor(rng.normal(0, 1, pos.shape))
    out = dict(xyz=pos.astype(F32), cell=cell.astype(F32), cutoff=2.5, w=w)
    for name, mk in PAIR_FORMS.items():
        model = mk()
        pp = PairPotentials(system, model, cutoff=2.5)
        q = torch.Tensor(pos).requires_grad_(True)
        pp._reset_topology(q.detach())
        r = compute_dis(q, pp.nbr_list, pp.offsets, pp.cell)
        u = model(r)
        U = pp(q)
        (g,) = torch.autograd.grad(U, q, create_graph=True)
        Fv = -g
        params = [p for p in model.parameters()]
        wF = (w * Fv).sum()
        grads = torch.autograd.grad(wF, [q] + params, allow_unused=True)
        Hw_neg = grads[0]          # d(w.F)/dq = -H w
        out[name + "_r"] = r.detach().reshape(-1)
        out[name + "_u"] = u.detach().reshape(-1)
        out[name + "_U"] = U.detach().reshape(1)
        out[name + "_F"] = Fv.detach()
        out[name + "_dwF_dq"] = Hw_neg
        out[name + "_dwF_dtheta"] = (torch.stack([x.reshape(()) for x in grads[1:]])
                                     if params else torch.zeros(0))
        out[name + "_theta"] = (torch.stack([p.detach().reshape(()) for p in params])
                                if params else torch.zeros(0))
        if name == "lj":
            out["nbr"] = pp.nbr_list
            out["offsets"] = pp.offsets
    save("pair_forms", **out)


# ------------------------------------------------------------------ G3-G5
def build_lj_sim(pos, cell, vel, model, T=1.0, Q=50.0, chains=5, cutoff=2.5, adjoint=True,
                 freq=1):
    system = make_system(pos, cell, vel=vel)
    pair = PairPotentials(system, model, cutoff=cutoff)
    stack = Stack({"pair": pair})
    integ = NoseHooverChain(stack, system, T=T, num_chains=chains, Q=Q, adjoint=adjoint,
                            topology_update_freq=freq)
    return system, integ


def g3_g4_g5():
    pos, cell, vel = lj_inputs(seed=0)
    model = P.LennardJones(sigma=1.0, epsilon=1.0)
    system, integ = build_lj_sim(pos, cell, vel, model)
    v0, q0, pv0 = integ.get_inital_states(wrap=True)
    pv_test = torch.Tensor([0.3, -0.2, 0.1, 0.05, -0.4])
    dv, dq, dpv = integ(torch.tensor(0.0), (v0.clone(), q0.clone(), pv_test))
    save("nhc_rhs", pos=pos.astype(F32), cell=cell.astype(F32), vel=vel.astype(F32),
         mass=system.get_masses().astype(F32), T=1.0, Q=50.0, chains=5, cutoff=2.5,
         sigma=1.0, epsilon=1.0, pv=pv_test, dv=dv.detach(), dq=dq.detach(), dpv=dpv.detach())

    # forward trajectory, 49 steps
    for kind, mdl, dt in [("lj", P.LennardJones(1.0, 1.0), 0.005),
                          ("exvol", P.ExcludedVolume(1.0, 1.0, 12), 0.01)]:
        system, integ = build_lj_sim(pos, cell, vel, mdl)
        y0 = tuple(integ.get_inital_states(wrap=True))
        t = torch.Tensor([dt * i for i in range(50)])
        with torch.no_grad():
            v_t, q_t, pv_t = odeint(integ, y0, t, method="NH_verlet")
        save("nhc_traj_" + kind, pos=pos.astype(F32), cell=cell.astype(F32), vel=vel.astype(F32),
             mass=system.get_masses().astype(F32), T=1.0, Q=50.0, chains=5, cutoff=2.5, dt=dt,
             v_t=v_t, q_t=q_t, pv_t=pv_t)

        # adjoint + rdf loss
        system, integ = build_lj_sim(pos, cell, vel, mdl)
        y0 = [s.clone().requires_grad_(True) for s in integ.get_inital_states(wrap=True)]
        obs = rdf(system, nbins=100, r_range=(0.75, 2.5))
        v_t, q_t, pv_t = odeint_adjoint(integ, tuple(y0), t, method="NH_verlet")
        count, bins, g = obs(q_t)
        target = torch.ones_like(g)
        loss = (g - target).pow(2).mean() + 0.01 * v_t[-1].pow(2).sum() + 0.1 * pv_t[-1].sum()
        loss.backward()
        th = list(mdl.parameters())
        save("nhc_adj_" + kind, pos=pos.astype(F32), cell=cell.astype(F32), vel=vel.astype(F32),
             mass=system.get_masses().astype(F32), T=1.0, Q=50.0, chains=5, cutoff=2.5, dt=dt,
             g=g.detach(), count=count.detach(), loss=loss.detach().reshape(1),
             grad_sigma=th[0].grad, grad_epsilon=th[1].grad,
             grad_v0=y0[0].grad, grad_q0=y0[1].grad, grad_pv0=y0[2].grad,
             q_last=q_t[-1].detach(), v_last=v_t[-1].detach())

    # short trajectory with stale neighbour list (topology_update_freq = 3)
    mdl = P.LennardJones(1.0, 1.0)
    system, integ = build_lj_sim(pos, cell, vel, mdl, freq=3)
    y0 = [s.clone().requires_grad_(True) for s in integ.get_inital_states(wrap=True)]
    t = torch.Tensor([0.005 * i for i in range(12)])
    v_t, q_t, pv_t = odeint_adjoint(integ, tuple(y0), t, method="NH_verlet")
    loss = q_t[::3].pow(2).mean() + v_t[-1].pow(2).mean()
    loss.backward()
    th = list(mdl.parameters())
    save("nhc_adj_freq3", pos=pos.astype(F32), cell=cell.astype(F32), vel=vel.astype(F32),
         mass=system.get_masses().astype(F32), T=1.0, Q=50.0, chains=5, cutoff=2.5, dt=0.005,
         v_t=v_t.detach(), q_t=q_t.detach(), pv_t=pv_t.detach(),
         grad_sigma=th[0].grad, grad_epsilon=th[1].grad,
         grad_v0=y0[0].grad, grad_q0=y0[1].grad, grad_pv0=y0[2].grad)


# ------------------------------------------------------------------ G6
def g6():
    pos, cell, vel = lj_inputs(seed=0)
    mdl = P.LennardJones(1.0, 1.0)
    system = make_system(pos, cell, vel=vel)
    pair = PairPotentials(system, mdl, cutoff=2.5)
    integ = NVE(Stack({"pair": pair}), system, adjoint=True)
    y0 = [s.clone().requires_grad_(True) for s in integ.get_inital_states(wrap=True)]
    t = torch.Tensor([0.005 * i for i in range(12)])
    v_t, q_t = odeint_adjoint(integ, tuple(y0), t, method="verlet")
    obs = rdf(system, nbins=100, r_range=(0.75, 2.5))
    _, _, g = obs(q_t)
    loss = g.pow(2).sum() + v_t[-1].pow(2).sum()
    loss.backward()
    th = list(mdl.parameters())
    save("nve_adj", pos=pos.astype(F32), cell=cell.astype(F32), vel=vel.astype(F32),
         mass=system.get_masses().astype(F32), cutoff=2.5, dt=0.005,
         v_t=v_t.detach(), q_t=q_t.detach(), g=g.detach(),
         grad_sigma=th[0].grad, grad_epsilon=th[1].grad, grad_v0=y0[0].grad, grad_q0=y0[1].grad)


# ------------------------------------------------------------------ G7
def g7():
    pos, cell, _ = lj_inputs(seed=7)
    system = make_system(pos, cell)
    rng = np.random.default_rng(8)
    frames = np.stack([np.mod(pos + rng.normal(0, 0.08, pos.shape), cell) for _ in range(4)])
    xyz = torch.Tensor(frames).requires_grad_(True)
    obs = rdf(system, nbins=100, r_range=(0.75, 2.5))
    count, bins, g = obs(xyz)
    wgt = torch.Tensor(np.random.default_rng(9).normal(0, 1, 100))
    (gx,) = torch.autograd.grad((g * wgt).sum(), xyz)
    out = dict(xyz=xyz.detach(), cell=cell.astype(F32), count=count.detach(), bins=bins, g=g.detach(),
               wgt=wgt, grad_xyz=gx, V=obs.V, vol_bins=obs.vol_bins)
    # species-selected rdf with explicit width, single frame
    idx_a = list(range(0, 108, 2))
    idx_b = list(range(1, 108, 2))
    obs2 = rdf(system, nbins=40, r_range=(0.5, 2.2), index_tuple=(idx_a, idx_b), width=0.07)
    x1 = torch.Tensor(frames[0]).requires_grad_(True)
    c2, b2, g2_ = obs2(x1)
    (gx2,) = torch.autograd.grad(g2_.pow(2).sum(), x1)
    out.update(idx_a=idx_a, idx_b=idx_b, sel_count=c2.detach(), sel_g=g2_.detach(), sel_bins=b2,
               sel_grad=gx2)
    save("rdf", **out)


# ------------------------------------------------------------------ G8 / G9
def diamond(size, a):
    basis = np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0],
                      [.25, .25, .25], [.25, .75, .75], [.75, .25, .75], [.75, .75, .25]])
    pts = []
    for i in range(size):
        for j in range(size):
            for k in range(size):
                for b in basis:
                    pts.append((np.array([i, j, k]) + b) * a)
    return np.array(pts), np.array([a * size] * 3)


def read_water():
    lines = open("/root/reference/data/water_init_64.xyz").read().splitlines()
    n = int(lines[0])
    L = float(lines[1].split('"')[1].split()[0])
    sym, xyz = [], []
    for ln in lines[2:2 + n]:
        s = ln.split()
        sym.append(8 if s[0] == "O" else 1)
        xyz.append([float(s[1]), float(s[2]), float(s[3])])
    return np.array(xyz), np.array(sym), np.array([L, L, L])


def g8_g9():
    torch.manual_seed(0)
    params = {"n_atom_basis": 32, "n_filters": 48, "n_gaussians": 16, "n_convolutions": 2,
              "cutoff": 5.0, "trainable_gauss": False}
    # CG water: 64 beads on a 2x2x2 diamond lattice, jittered
    a = 6.2148
    pos, cell = diamond(2, a)
    rng = np.random.default_rng(10)
    pos = np.mod(pos + rng.normal(0, 0.3, pos.shape), cell).astype(F32).astype(np.float64)
    numbers = np.full(len(pos), 8)
    masses = np.full(len(pos), 18.01528)
    vel = rng.normal(0, 0.01, pos.shape).astype(F32).astype(np.float64)
    system = make_system(pos, cell, numbers=numbers, masses=masses, vel=vel)
    net = SchNet(params)
    gnn = GNNPotentials(system, net, cutoff=5.0)
    q = torch.Tensor(pos).requires_grad_(True)
    gnn._reset_topology(q.detach())
    U = gnn(q)
    (gq,) = torch.autograd.grad(U.sum(), q, create_graph=True)
    Fv = -gq
    w = torch.Tensor(rng.normal(0, 1, pos.shape))
    plist = list(net.parameters())
    grads = torch.autograd.grad((w * Fv).sum(), [q] + plist, allow_unused=True)
    flat = torch.cat([(g if g is not None else torch.zeros_like(p)).reshape(-1)
                      for g, p in zip(grads[1:], plist)])
    sd = {"sd__" + k: v for k, v in net.state_dict().items()}
    save("schnet_cg64", pos=pos.astype(F32), cell=cell.astype(F32), numbers=numbers,
         masses=masses.astype(F32), vel=vel.astype(F32),
         n_atom_basis=32, n_filters=48, n_gaussians=16, n_convolutions=2, cutoff=5.0,
         nbr=gnn.inputs["nbr_list"], offsets=gnn.inputs["offsets"],
         U=U.detach().reshape(-1), F=Fv.detach(), w=w, dwF_dq=grads[0], dwF_dtheta=flat, **sd)

    # all-atom water geometry (192 atoms, O/H species)
    wpos, wnum, wcell = read_water()
    wpos = np.mod(wpos, wcell).astype(F32).astype(np.float64)
    wmass = np.where(wnum == 8, 15.999, 1.008)
    wsys = make_system(wpos, wcell, numbers=wnum, masses=wmass)
    torch.manual_seed(1)
    net2 = SchNet(params)
    gnn2 = GNNPotentials(wsys, net2, cutoff=5.0)
    q2 = torch.Tensor(wpos).requires_grad_(True)
    gnn2._reset_topology(q2.detach())
    U2 = gnn2(q2)
    (g2q,) = torch.autograd.grad(U2.sum(), q2)
    sd2 = {"sd__" + k: v for k, v in net2.state_dict().items()}
    save("schnet_water192", pos=wpos.astype(F32), cell=wcell.astype(F32), numbers=wnum,
         masses=wmass.astype(F32), n_atom_basis=32, n_filters=48, n_gaussians=16,
         n_convolutions=2, cutoff=5.0, nbr=gnn2.inputs["nbr_list"],
         offsets=gnn2.inputs["offsets"], U=U2.detach().reshape(-1), F=-g2q, **sd2)

    # G9: Stack(GNN + ExcludedVolume prior) NHC trajectory + adjoint
    system = make_system(pos, cell, numbers=numbers, masses=masses, vel=vel)
    torch.manual_seed(0)
    net = SchNet(params)
    gnn = GNNPotentials(system, net, cutoff=5.0)
    prior_model = P.ExcludedVolume(sigma=2.6, epsilon=0.01, power=12)
    prior = PairPotentials(system, prior_model, cutoff=5.0)
    stack = Stack({"gnn": gnn, "prior": prior})
    kT = 298.0 * 8.617330337217213e-05
    integ = NoseHooverChain(stack, system, T=kT, num_chains=5, Q=50.0, adjoint=True)
    y0 = [s.clone().requires_grad_(True) for s in integ.get_inital_states(wrap=True)]
    dt = 1.0 * 0.09822694788464063
    t = torch.Tensor([dt * i for i in range(11)])
    v_t, q_t, pv_t = odeint_adjoint(integ, tuple(y0), t, method="NH_verlet")
    obs = rdf(system, nbins=40, r_range=(2.0, 5.5))
    _, _, g = obs(q_t[::2])
    loss = g.pow(2).mean() + q_t[-1].pow(2).mean() * 1e-3
    loss.backward()
    plist = list(integ.parameters())
    flatg = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                       for p in plist])
    names = [n for n, _ in integ.named_parameters()]
    save("gnn_traj", pos=pos.astype(F32), cell=cell.astype(F32), numbers=numbers,
         masses=masses.astype(F32), vel=vel.astype(F32), T=kT, Q=50.0, chains=5, dt=dt,
         cutoff=5.0, prior_sigma=2.6, prior_epsilon=0.01,
         n_atom_basis=32, n_filters=48, n_gaussians=16, n_convolutions=2,
         v_t=v_t.detach(), q_t=q_t.detach(), pv_t=pv_t.detach(), g=g.detach(),
         loss=loss.detach().reshape(1), grad_flat=flatg, param_names=np.array(names),
         grad_q0=y0[1].grad, grad_v0=y0[0].grad, **sd)


# ------------------------------------------------------------------ G14 / G15
def g14_g15():
    """G14: BASELINE config #3 -- the 192-atom all-atom water box with a SchNet (A128/F128/G32/3 conv, cutoff 5) +
    ExcludedVolume prior, NHC trajectory and adjoint of an RDF loss.  G15: energy / force / H.w / d(w.F)/dtheta of a
    SchNet with the widths of BASELINE config #5 (A64/F128/G30/2 conv, cutoff 6) on the 64-bead CG box."""
    from torchmd.thermo import Temperature  # noqa: F401  (import check only)
    wpos, wnum, wcell = read_water()
    wpos = np.mod(wpos, wcell).astype(F32).astype(np.float64)
    wmass = np.where(wnum == 8, 15.999, 1.008)
    rng = np.random.default_rng(14)
    kT = 298.0 * 8.617330337217213e-05
    wvel = (rng.normal(0, 1, wpos.shape) * np.sqrt(kT / wmass)[:, None]).astype(F32).astype(np.float64)
    params = {"n_atom_basis": 128, "n_filters": 128, "n_gaussians": 32, "n_convolutions": 3,
              "cutoff": 5.0, "trainable_gauss": False}
    system = make_system(wpos, wcell, numbers=wnum, masses=wmass, vel=wvel)
    torch.manual_seed(0)
    net = SchNet(params)
    with torch.no_grad():                      # random-init forces are O(100 eV/A): scale the readout (stable dynamics)
        net.atomwisereadout.readout["energy"][2].weight.mul_(0.02)
    gnn = GNNPotentials(system, net, cutoff=5.0)
    prior_model = P.ExcludedVolume(sigma=0.9, epsilon=0.01, power=12)
    prior = PairPotentials(system, prior_model, cutoff=5.0)
    integ = NoseHooverChain(Stack({"gnn": gnn, "prior": prior}), system, T=kT, num_chains=5, Q=50.0, adjoint=True)
    y0 = [s_.clone().requires_grad_(True) for s_ in integ.get_inital_states(wrap=True)]
    dt = 0.25 * 0.09822694788464063
    t = torch.Tensor([dt * i for i in range(9)])
    v_t, q_t, pv_t = odeint_adjoint(integ, tuple(y0), t, method="NH_verlet")
    O_idx, H_idx = np.where(wnum == 8)[0].tolist(), np.where(wnum == 1)[0].tolist()
    obs = rdf(system, nbins=40, r_range=(0.6, 5.0), index_tuple=(O_idx, H_idx))
    _, _, g = obs(q_t[::2])
    loss = g.pow(2).mean() + q_t[-1].pow(2).mean() * 1e-3 + v_t[-1].pow(2).sum() * 1e-2
    loss.backward()
    plist = list(integ.parameters())
    flatg = torch.cat([(p_.grad if p_.grad is not None else torch.zeros_like(p_)).reshape(-1) for p_ in plist])
    names = [n for n, _ in integ.named_parameters()]
    sd = {"sd__" + k: v for k, v in net.state_dict().items()}
    save("gnn_traj_water192", pos=wpos.astype(F32), cell=wcell.astype(F32), numbers=wnum, masses=wmass.astype(F32),
         vel=wvel.astype(F32), T=kT, Q=50.0, chains=5, dt=dt, cutoff=5.0, prior_sigma=0.9, prior_epsilon=0.01,
         n_atom_basis=128, n_filters=128, n_gaussians=32, n_convolutions=3,
         v_t=v_t.detach(), q_t=q_t.detach(), pv_t=pv_t.detach(), g=g.detach(), loss=loss.detach().reshape(1),
         grad_flat=flatg, param_names=np.array(names), grad_q0=y0[1].grad, grad_v0=y0[0].grad,
         idx_O=np.array(O_idx), idx_H=np.array(H_idx), **sd)

    # G15: bench widths on the CG box
    a = 6.2148
    pos, cell = diamond(2, a)
    rng = np.random.default_rng(15)
    pos = np.mod(pos + rng.normal(0, 0.3, pos.shape), cell).astype(F32).astype(np.float64)
    numbers = np.full(len(pos), 8)
    masses = np.full(len(pos), 18.01528)
    system = make_system(pos, cell, numbers=numbers, masses=masses)
    params = {"n_atom_basis": 64, "n_filters": 128, "n_gaussians": 30, "n_convolutions": 2,
              "cutoff": 6.0, "trainable_gauss": False}
    torch.manual_seed(5)
    net = SchNet(params)
    gnn = GNNPotentials(system, net, cutoff=6.0)
    q = torch.Tensor(pos).requires_grad_(True)
    gnn._reset_topology(q.detach())
    U = gnn(q)
    (gq,) = torch.autograd.grad(U.sum(), q, create_graph=True)
    w = torch.Tensor(rng.normal(0, 1, pos.shape))
    plist = list(net.parameters())
    grads = torch.autograd.grad((w * -gq).sum(), [q] + plist, allow_unused=True)
    flat = torch.cat([(g_ if g_ is not None else torch.zeros_like(p_)).reshape(-1) for g_, p_ in zip(grads[1:], plist)])
    sd = {"sd__" + k: v for k, v in net.state_dict().items()}
    save("schnet_cg64_wide", pos=pos.astype(F32), cell=cell.astype(F32), numbers=numbers, masses=masses.astype(F32),
         n_atom_basis=64, n_filters=128, n_gaussians=30, n_convolutions=2, cutoff=6.0,
         nbr=gnn.inputs["nbr_list"], offsets=gnn.inputs["offsets"], U=U.detach().reshape(-1), F=-gq.detach(), w=w,
         dwF_dq=grads[0], dwF_dtheta=flat, **sd)


# ------------------------------------------------------------------ G16
def g16():
    """vacf (torchmd/observable.py:153-163) and Temperature (torchmd/thermo.py:57-66) on a seeded velocity
    trajectory, with the gradient of a weighted sum w.r.t. the velocities."""
    from torchmd.observable import vacf
    from torchmd.thermo import Temperature
    rng = np.random.default_rng(16)
    pos, cell, vel = lj_inputs(seed=16)
    masses = rng.uniform(0.8, 3.0, len(pos)).astype(F32).astype(np.float64)
    system = make_system(pos, cell, masses=masses, vel=vel)
    v_t = torch.Tensor(rng.normal(0, 1, (30, len(pos), 3)).astype(F32)).requires_grad_(True)
    wgt = torch.Tensor(rng.normal(0, 1, 12).astype(F32))
    c = vacf(system, t_range=12)(v_t)
    (gv,) = torch.autograd.grad((c * wgt).sum(), v_t)
    temp = Temperature(system)
    v1 = torch.Tensor(vel).requires_grad_(True)
    T1 = temp(v1)
    (gT,) = torch.autograd.grad(T1, v1)
    Tt = torch.stack([temp(v_t[k]) for k in range(v_t.shape[0])])
    save("vacf_temp", pos=pos.astype(F32), cell=cell.astype(F32), masses=masses.astype(F32), vel=vel.astype(F32),
         v_t=v_t.detach(), wgt=wgt, vacf=c.detach(), vacf_grad=gv, T_single=T1.detach().reshape(1), T_grad=gT,
         T_frames=Tt.detach())


# ------------------------------------------------------------------ G10
def g10():
    pos, cell, vel = lj_inputs(seed=0)
    mdl = P.ExcludedVolume(1.0, 1.0, 12)
    system, integ = build_lj_sim(pos, cell, vel, mdl)
    sim = Simulations(system, integ, wrap=True, method="NH_verlet")
    v_t, q_t, pv_t = sim.simulate(steps=20, frequency=10, dt=0.01)
    obs = rdf(system, nbins=100, r_range=(0.75, 2.5))
    _, _, g = obs(q_t)
    g.sum().backward()
    th = list(mdl.parameters())
    save("sim_2epoch", pos=pos.astype(F32), cell=cell.astype(F32), vel=vel.astype(F32),
         mass=system.get_masses().astype(F32), T=1.0, Q=50.0, chains=5, cutoff=2.5, dt=0.01,
         v_t=v_t.detach(), q_t=q_t.detach(), pv_t=pv_t.detach(),
         log_positions=np.stack(sim.log["positions"]), log_velocities=np.stack(sim.log["velocities"]),
         log_baths=np.stack(sim.log["baths"]), g=g.detach(),
         grad_sigma=th[0].grad, grad_epsilon=th[1].grad,
         sys_positions=system.get_positions(), sys_velocities=system.get_velocities())


# ------------------------------------------------------------------ G11
def g11():
    """pairMLP / TpairMLP: the per-pair MLP potentials of the LJ-fitting scripts."""
    pos, cell, vel = lj_inputs(seed=3)
    kw = dict(n_gauss=16, r_start=0.0, r_end=2.5, n_layers=1, n_width=24)
    out = dict(pos=pos.astype(F32), cell=cell.astype(F32), vel=vel.astype(F32), cutoff=2.5, T=1.0, Q=50.0,
               chains=5, dt=0.005, n_gauss=16, n_layers=1, n_width=24, r_end=2.5)
    r = torch.linspace(0.6, 2.4, 37)[:, None]
    for tag, nonlinear, res in [("elu", "ELU", False), ("tanh_res", "Tanh", True)]:
        torch.manual_seed(11)
        mlp = P.pairMLP(nonlinear=nonlinear, res=res, **kw)
        for k, v in mlp.state_dict().items():
            out["%s_sd_%s" % (tag, k)] = v.clone()
        out[tag + "_r"] = r[:, 0]
        out[tag + "_u"] = mlp(r)[:, 0].detach()
    # energies / forces through PairPotentials, and the temperature-dependent pair model
    torch.manual_seed(11)
    mlp = P.pairMLP(nonlinear="ELU", res=False, **kw)
    system = make_system(pos, cell, vel=vel)
    pp = PairPotentials(system, mlp, cutoff=2.5)
    q = torch.Tensor(pos).requires_grad_(True)
    u = pp(q)
    (gq,) = torch.autograd.grad(u, q)
    out["pp_energy"], out["pp_force"] = u.detach().reshape(1), -gq
    torch.manual_seed(12)
    tm = P.TpairMLP(nonlinear="ELU", res=False, **kw)
    for k, v in tm.state_dict().items():
        out["t_sd_" + k] = v.clone()
    tp = TPairPotentials(system, tm, T=150.0, cutoff=2.5)
    q = torch.Tensor(pos).requires_grad_(True)
    ut = tp(q)
    (gqt,) = torch.autograd.grad(ut, q)
    out["tp_T"], out["tp_energy"], out["tp_force"] = 150.0, ut.detach().reshape(1), -gqt
    # Stack(pairMLP + LJFamily prior) NHC trajectory and adjoint of an RDF loss (scripts/fit_rdf_pair.py:355-368)
    torch.manual_seed(11)
    mlp = P.pairMLP(nonlinear="ELU", res=False, **kw)
    prior = P.LJFamily(epsilon=2.0, sigma=0.9, rep_pow=6, attr_pow=3)
    system = make_system(pos, cell, vel=vel)
    stack = Stack({"pairnn": PairPotentials(system, mlp, cutoff=2.5), "pair": PairPotentials(system, prior, cutoff=2.5)})
    integ = NoseHooverChain(stack, system, T=1.0, num_chains=5, Q=50.0, adjoint=True)
    y0 = [s_.clone().requires_grad_(True) for s_ in integ.get_inital_states(wrap=True)]
    t = torch.Tensor([0.005 * i for i in range(9)])
    v_t, q_t, pv_t = odeint_adjoint(integ, tuple(y0), t, method="NH_verlet")
    obs = rdf(system, nbins=60, r_range=(0.75, 2.4))
    _, _, g = obs(q_t)
    loss = (g - 1).pow(2).mean() + 0.01 * v_t[-1].pow(2).sum()
    loss.backward()
    out.update(v_t=v_t.detach(), q_t=q_t.detach(), pv_t=pv_t.detach(), g=g.detach(), loss=loss.detach().reshape(1),
               grad_v0=y0[0].grad, grad_q0=y0[1].grad,
               grad_mlp=torch.cat([p_.grad.reshape(-1) for p_ in mlp.parameters()]),
               grad_prior=torch.cat([p_.grad.reshape(-1) for p_ in prior.parameters()]),
               prior_sigma=0.9, prior_epsilon=2.0, mass=system.get_masses().astype(F32))
    save("pair_mlp", **out)


# ------------------------------------------------------------------ G12
def g12():
    """BondPotentials / AnglePotentials: a 24-bead chain across the periodic boundary."""
    rng = np.random.default_rng(7)
    n, L = 24, 6.0
    steps = rng.normal(0, 1, (n, 3))
    steps = 1.1 * steps / np.linalg.norm(steps, axis=1)[:, None]
    pos = np.mod(np.cumsum(steps, 0) + 2.5, L)
    cell = np.array([L, L, L])
    system = make_system(pos, cell)
    bonds = torch.LongTensor([[i, i + 1] for i in range(n - 1)])
    angles = torch.LongTensor([[i, i + 1, i + 2] for i in range(n - 2)])
    out = dict(pos=pos.astype(F32), cell=cell.astype(F32), bonds=bonds.numpy(), angles=angles.numpy(),
               k_bond=3.0, ro=1.21, k_angle=2.0, theta0=1.9)
    for tag, mod in [("bond", BondPotentials(system, bonds, 3.0, 1.21)), ("angle", AnglePotentials(system, angles, 2.0, 1.9))]:
        q = torch.Tensor(pos).requires_grad_(True)
        u = mod(q)
        (gq,) = torch.autograd.grad(u, q)
        out[tag + "_energy"], out[tag + "_force"] = u.detach().reshape(1), -gq
    save("bonded", **out)


# ------------------------------------------------------------------ G18
def g18():
    """f4 in the adjoint.  (a) bonded_hvp: H.w of BondPotentials / AnglePotentials by double autograd --
    what the reference's adjoint derives at sovlers.py:229-233 -- on the chain of G12.  (b) fold_traj: the polymer Stack of
    demo/fold.py:131-161 -- {'gnn': GNNPotentials, 'prior': BondPotentials, 'pair': ExcludedVolume(power 10, cutoff 2.5) with the
    bonded pairs excluded} -- and the same without the GNN, NoseHooverChain(Q = 50, 5 chains) trajectory + adjoint.
    (AnglePotentials has no _reset_topology in the reference and cannot be a Stack member there: its golden is (a).)"""
    rng = np.random.default_rng(7)
    n, L = 24, 6.0
    steps = rng.normal(0, 1, (n, 3))
    steps = 1.1 * steps / np.linalg.norm(steps, axis=1)[:, None]
    pos = np.mod(np.cumsum(steps, 0) + 2.5, L).astype(F32).astype(np.float64)
    cell = np.array([L, L, L])
    bonds = torch.LongTensor([[i, i + 1] for i in range(n - 1)])
    angles = torch.LongTensor([[i, i + 1, i + 2] for i in range(n - 2)])
    system = make_system(pos, cell)
    rng = np.random.default_rng(18)
    w = rng.normal(0, 1, pos.shape).astype(F32)
    out = dict(pos=pos.astype(F32), cell=cell.astype(F32), bonds=bonds.numpy(), angles=angles.numpy(), w=w,
               k_bond=3.0, ro=1.21, k_angle=2.0, theta0=1.9)
    for tag, mod in [("bond", BondPotentials(system, bonds, 3.0, 1.21)), ("angle", AnglePotentials(system, angles, 2.0, 1.9))]:
        q = torch.Tensor(pos).requires_grad_(True)
        u = mod(q)
        (gq,) = torch.autograd.grad(u, q, create_graph=True)
        (hw,) = torch.autograd.grad((gq * torch.Tensor(w)).sum(), q)
        out[tag + "_energy"], out[tag + "_force"], out[tag + "_hw"] = u.detach().reshape(1), -gq.detach(), hw
    save("bonded_hvp", **out)

    # (b) a self-avoiding chain across the periodic boundary: no non-bonded pair closer than 1.0 (the random walk above has
    # beads on top of each other, fine for the bonded terms alone but not under an excluded-volume term)
    def min_dist(p, others):
        d = others - p
        d -= L * np.round(d / L)
        return np.sqrt((d ** 2).sum(1)).min()
    chain = [np.array([2.5, 2.5, 2.5])]
    while len(chain) < n:
        st = rng.normal(0, 1, 3)
        cand = chain[-1] + 1.1 * st / np.linalg.norm(st)
        if len(chain) < 2 or min_dist(cand, np.array(chain[:-1])) > 1.0:
            chain.append(cand)
    pos = np.mod(np.array(chain), L).astype(F32).astype(np.float64)
    T_, dt, nsteps = 0.5, 0.005, 12
    masses = np.full(n, 1.008)
    vel = rng.normal(0, np.sqrt(T_ / 1.008), pos.shape).astype(F32).astype(np.float64)
    params = {"n_atom_basis": 32, "n_filters": 48, "n_gaussians": 16, "n_convolutions": 2, "cutoff": 2.5,
              "trainable_gauss": False}
    out = dict(pos=pos.astype(F32), cell=cell.astype(F32), vel=vel.astype(F32), masses=masses.astype(F32),
               numbers=np.ones(n, dtype=np.int64), bonds=bonds.numpy(), k_bond=3.0, ro=1.21, T=T_, Q=50.0, chains=5, dt=dt,
               pair_cutoff=2.5, pair_sigma=0.9, pair_epsilon=0.5, pair_power=10,
               **{k: v for k, v in params.items() if k != "trainable_gauss"})
    for tag in ("fold", "pairbond"):
        system = make_system(pos, cell, numbers=np.ones(n, dtype=np.int64), masses=masses, vel=vel)
        bond = BondPotentials(system, bonds, 3.0, 1.21)
        pair_model = P.ExcludedVolume(sigma=0.9, epsilon=0.5, power=10)
        pair = PairPotentials(system, pair_model, cutoff=2.5, ex_pairs=bonds)        # demo/fold.py:150-155
        if tag == "fold":
            torch.manual_seed(18)
            net = SchNet(params)
            with torch.no_grad():                      # (random-init SchNet forces are O(100): keep the 12 steps gentle)
                net.atomwisereadout.readout["energy"][2].weight.mul_(0.05)
            out.update({"sd__" + k: v.detach().clone() for k, v in net.state_dict().items()})
            gnn = GNNPotentials(system, net, cutoff=2.5)
            stack = Stack({"gnn": gnn, "prior": bond, "pair": pair})                 # demo/fold.py:157-161
        else:
            stack = Stack({"pair": pair, "bond": bond})
        integ = NoseHooverChain(stack, system, T=T_, num_chains=5, Q=50.0, adjoint=True)
        y0 = [s_.clone().requires_grad_(True) for s_ in integ.get_inital_states(wrap=True)]
        t = torch.Tensor([dt * i for i in range(nsteps + 1)])
        v_t, q_t, pv_t = odeint_adjoint(integ, tuple(y0), t, method="NH_verlet")
        loss = q_t[::3].pow(2).mean() * 1e-2 + v_t[-1].pow(2).mean() + pv_t[-1].sum() * 1e-2
        loss.backward()
        plist = list(integ.parameters())
        flatg = torch.cat([(p_.grad if p_.grad is not None else torch.zeros_like(p_)).reshape(-1) for p_ in plist])
        out.update({tag + "_v_t": v_t.detach(), tag + "_q_t": q_t.detach(), tag + "_pv_t": pv_t.detach(),
                    tag + "_loss": loss.detach().reshape(1), tag + "_grad_flat": flatg,
                    tag + "_param_names": np.array([nm for nm, _ in integ.named_parameters()]),
                    tag + "_grad_q0": y0[1].grad, tag + "_grad_v0": y0[0].grad, tag + "_grad_pv0": y0[2].grad})
    save("fold_traj", **out)


# ------------------------------------------------------------------ G13
def g13():
    """get_exp_rdf (scripts/data.py:11-31): target g(r) on the observable's grid from tabulated data.
    scripts/data.py cannot be imported as a module here (its module-level tables need torchcubicspline), so
    only that function is compiled out of the reference file and run."""
    import ast
    from scipy import interpolate
    from torchmd.observable import generate_vol_bins
    src = open("/root/reference/scripts/data.py").read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "get_exp_rdf"]
    ns = dict(torch=torch, np=np, interpolate=interpolate, generate_vol_bins=generate_vol_bins)
    exec(compile(ast.Module(body=fn, type_ignores=[]), "scripts/data.py", "exec"), ns)
    r = np.linspace(0.5, 8, 200)
    g = 1 + np.exp(-(r - 3) ** 2) * np.cos(4 * r)
    x, g_obs = ns["get_exp_rdf"](np.stack([r, g]).T, 33, (1.0, 7.5), "cpu")
    save("exp_rdf", r=r.astype(F32), g=g.astype(F32), x=x.astype(F32), g_obs=g_obs)


# ------------------------------------------------------------------ G17
def g17():
    """SchNet at the wide settings of the reference's search space (demo/fit_rdf_gnn.py:16-19, 127-134: n_atom_basis =
    n_filters = 256, n_gaussians = int(cutoff // gaussian_width) = 41) on the 64-bead CG box: U, F, H.w, d(w.F)/dtheta."""
    a = 6.2148
    pos, cell = diamond(2, a)
    rng = np.random.default_rng(17)
    pos = np.mod(pos + rng.normal(0, 0.3, pos.shape), cell).astype(F32).astype(np.float64)
    numbers = np.full(len(pos), 8)
    masses = np.full(len(pos), 18.01528)
    system = make_system(pos, cell, numbers=numbers, masses=masses)
    params = {"n_atom_basis": 256, "n_filters": 256, "n_gaussians": 41, "n_convolutions": 2,
              "cutoff": 6.0, "trainable_gauss": False}
    torch.manual_seed(17)
    net = SchNet(params)
    gnn = GNNPotentials(system, net, cutoff=6.0)
    q = torch.Tensor(pos).requires_grad_(True)
    gnn._reset_topology(q.detach())
    U = gnn(q)
    (gq,) = torch.autograd.grad(U.sum(), q, create_graph=True)
    w = torch.Tensor(rng.normal(0, 1, pos.shape))
    plist = list(net.parameters())
    grads = torch.autograd.grad((w * -gq).sum(), [q] + plist, allow_unused=True)
    flat = torch.cat([(g_ if g_ is not None else torch.zeros_like(p_)).reshape(-1) for g_, p_ in zip(grads[1:], plist)])
    # the embedding table has 100 rows of which one (Z = 8) is used: keep that row only (the others are never read and
    # their gradient is zero -- the test rebuilds the table around it)
    sd = {"sd__" + k: v for k, v in net.state_dict().items() if k != "atom_embed.weight"}
    save("schnet_cg64_a256", pos=pos.astype(F32), cell=cell.astype(F32), numbers=numbers, masses=masses.astype(F32),
         n_atom_basis=256, n_filters=256, n_gaussians=41, n_convolutions=2, cutoff=6.0,
         nbr=gnn.inputs["nbr_list"], offsets=gnn.inputs["offsets"], U=U.detach().reshape(-1), F=-gq.detach(), w=w,
         dwF_dq=grads[0], dwF_dtheta=flat, embed_row8=net.state_dict()["atom_embed.weight"][8], **sd)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g345", "g6", "g7", "g89", "g10", "g11", "g12", "g13", "g1415", "g16", "g17", "g18"]
    table = {"g1": g1, "g2": g2, "g345": g3_g4_g5, "g6": g6, "g7": g7, "g89": g8_g9, "g10": g10, "g11": g11,
             "g12": g12, "g13": g13, "g1415": g14_g15, "g16": g16, "g17": g17, "g18": g18}
    for w in which:
        table[w]()
