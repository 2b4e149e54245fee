#!/usr/bin/env python
"""Learn a neural pair potential from a target RDF -- the loop of the reference's scripts/fit_rdf_pair.py
(:340-520: pairMLP + LJFamily prior in a Stack, NoseHooverChain, Simulations epochs continuing from the
last frame, JS-divergence + MSE loss on g(r), Adam + ReduceLROnPlateau) with R replicas stacked in one state
(System.replicate).  The pair energy is tabulated once per epoch and the whole epoch -- forward trajectory
and adjoint -- runs in the fused HIP kernels (MDG_PAIR_TABLE); autograd carries the table gradient into
the MLP.

    python examples/fit_rdf_pairmlp.py --replicas 256 --epochs 40
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def JS_rdf(g_obs, g, e0=1e-4):
    """Jensen-Shannon-style divergence between two RDFs (reference demo/fit_rdf_gnn.py:36-42)."""
    g_m = 0.5 * (g_obs + g)
    js = (-(g_obs + e0) * (torch.log(g_m + e0) - torch.log(g_obs + e0))).mean()
    return js + (-(g + e0) * (torch.log(g_m + e0) - torch.log(g + e0))).mean()


def make_system(R, dev, seed):
    from mdgrad_amd.system import System, FaceCenteredCubic
    rng = np.random.default_rng(seed)
    system = System(FaceCenteredCubic("H", (3, 3, 3), 1.6), device=dev)
    if R > 1:
        system = system.replicate(R)
    L = 4.8
    system.set_positions(np.mod(system.get_positions() + rng.uniform(-0.05, 0.05, (len(system), 3)), L))
    system.set_velocities(rng.normal(0, np.sqrt(1.0 / 1.008), (len(system), 3)))
    return system


def target_rdf(R, frames, dev, obs_kw):
    """RDF of the 'experimental' liquid: Lennard-Jones(1, 1) at T = 1."""
    from mdgrad_amd import potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain, Simulations
    from mdgrad_amd.observable import rdf
    system = make_system(R, dev, 999)
    integ = NoseHooverChain(Stack({"lj": PairPotentials(system, P.LennardJones(1.0, 1.0), cutoff=2.5)}), system,
                            T=1.0, num_chains=5, Q=50.0).to(dev)
    sim = Simulations(system, integ)
    with torch.no_grad():
        sim.simulate(steps=2 * frames, frequency=frames, dt=0.005)           # equilibrate one epoch, sample one
        q_t = sim.simulate(steps=frames, frequency=frames, dt=0.005)[1]
    return rdf(system, **obs_kw)(q_t)[2].detach()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--replicas", type=int, default=256)
    ap.add_argument("--epochs", type=int, default=40)
    ap.add_argument("--frames", type=int, default=50)
    ap.add_argument("--lr", type=float, default=3e-3)
    ap.add_argument("--mse-weight", type=float, default=1.0)
    ap.add_argument("--warmup", type=int, default=6, help="epochs of plain MD before the first update (equilibration)")
    args = ap.parse_args(argv)
    from mdgrad_amd import potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain, Simulations
    from mdgrad_amd.observable import rdf
    dev = "cuda:0"
    obs_kw = dict(nbins=100, r_range=(0.75, 2.5))
    g_target = target_rdf(args.replicas, args.frames, dev, obs_kw)

    system = make_system(args.replicas, dev, 0)
    torch.manual_seed(0)
    mlp = P.pairMLP(n_gauss=25, r_start=0.0, r_end=2.5, n_layers=2, n_width=64, nonlinear="ELU")
    with torch.no_grad():
        mlp.layers[-1].weight.mul_(0.1)                                     # start close to the prior alone
    prior = P.LJFamily(epsilon=2.0, sigma=0.9, rep_pow=6, attr_pow=3)      # soft repulsive prior (fit_rdf_pair.py:356)
    for p in prior.parameters():
        p.requires_grad_(False)
    model = Stack({"pairnn": PairPotentials(system, mlp, cutoff=2.5), "pair": PairPotentials(system, prior, cutoff=2.5)})
    integ = NoseHooverChain(model, system, T=1.0, num_chains=5, Q=50.0).to(dev)
    assert integ.fused_spec("NH_verlet") is not None, "expected the tabulated fused path"
    sim = Simulations(system, integ)
    obs = rdf(system, **obs_kw)
    opt = torch.optim.Adam(mlp.parameters(), lr=args.lr)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, "min", factor=0.5, patience=8, min_lr=1e-5)
    hist = []
    with torch.no_grad():                      # the starting liquid of the untrained model, not the lattice
        sim.simulate(steps=args.warmup * args.frames, frequency=args.frames, dt=0.005)
    for epoch in range(args.epochs):
        opt.zero_grad()
        v_t, q_t, pv_t = sim.simulate(steps=args.frames, frequency=args.frames, dt=0.005)
        g = obs(q_t[10:])[2]
        loss_js, loss_mse = JS_rdf(g_target, g), (g - g_target).pow(2).mean()
        loss = loss_js + args.mse_weight * loss_mse
        loss.backward()
        opt.step()
        sched.step(float(loss.detach()))
        hist.append((float(loss), float(loss_js), float(loss_mse)))
        print("epoch %3d  loss %.5f  (JS %.5f, MSE %.5f)  lr %.1e" % (epoch, *hist[-1], opt.param_groups[0]["lr"]),
              flush=True)
    return hist


if __name__ == "__main__":
    main()
