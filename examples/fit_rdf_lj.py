#!/usr/bin/env python
"""Fit Lennard-Jones (sigma, epsilon) to a target RDF by back-propagating through the MD
trajectory -- the inner loop of the reference's scripts/fit_rdf_pair.py / demo/fit_rdf_gnn.py
(simulate -> rdf -> loss -> backward -> Adam), with R replica trajectories per step running as one
fused forward and one fused adjoint launch, and (on N GPUs) one gradient all-reduce per step.

    python examples/fit_rdf_lj.py --replicas 64 --epochs 30
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build(sigma, epsilon, dev, seed=0):
    from mdgrad_amd import potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    from mdgrad_amd.system import System, FaceCenteredCubic
    system = System(FaceCenteredCubic("H", (3, 3, 3), 1.6), device=dev)
    system.set_temperature(1.0, rng=np.random.default_rng(seed))
    model = P.LennardJones(sigma, epsilon)
    integ = NoseHooverChain(Stack({"pair": PairPotentials(system, model, cutoff=2.5)}), system, T=1.0,
                            num_chains=5, Q=50.0).to(dev)
    return system, model, integ


def trajectories(system, integ, R, frames, dt, seed, dev):
    """R replicas (jittered lattice, Maxwell-Boltzmann velocities) through the fused kernels."""
    from mdgrad_amd import ops
    rng = np.random.default_rng(seed)
    lat = system.get_positions()
    pos = np.mod(lat[None] + rng.uniform(-0.05, 0.05, (R,) + lat.shape), 4.8).astype(np.float32)
    vel = rng.normal(0, np.sqrt(1.0 / 1.008), pos.shape).astype(np.float32)
    t = torch.Tensor([dt * i for i in range(frames)]).to(dev)
    integ.fuse_observables = True            # opt-in: the loss only reads g(r), nothing differentiates w.r.t. q_t itself
    spec = integ.fused_spec("NH_verlet")
    # (fused_traj = FusedTrajFn + the bookkeeping that lets `rdf` ride inside the trajectory kernels from the second
    #  epoch on when there are >= 1 024 replicas: the observable registers itself the first time it sees q_t)
    return ops.fused_traj(torch.from_numpy(vel).to(dev), torch.from_numpy(pos).to(dev),
                          torch.zeros(R, 5, device=dev), t, spec.flat_params(), spec)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--replicas", type=int, default=64)
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--lr", type=float, default=0.01)
    args = ap.parse_args(argv)
    from mdgrad_amd import dist as mdist
    from mdgrad_amd.observable import rdf
    rank, world, dev = mdist.init()
    # target: RDF of the "true" liquid (sigma = 1.0, eps = 1.0)
    system, _, integ_true = build(1.0, 1.0, dev)
    obs = rdf(system, nbins=100, r_range=(0.75, 2.5))
    with torch.no_grad():
        q_true = trajectories(system, integ_true, args.replicas, args.frames, 0.005, 999, dev)[1]
        g_target = obs(q_true[:, 20:])[2]
    # start from a wrong potential
    system, model, integ = build(0.92, 0.8, dev)
    opt = torch.optim.Adam(integ.parameters(), lr=args.lr)
    hist = []
    for epoch in range(args.epochs):
        opt.zero_grad()
        v_t, q_t, pv_t = trajectories(system, integ, args.replicas, args.frames, 0.005, 100 + epoch * world + rank, dev)
        g = obs(q_t[:, 20:])[2]
        loss = (g - g_target).pow(2).mean()
        loss.backward()
        mdist.all_reduce_grads(integ.parameters(), average=True)
        opt.step()
        hist.append((float(loss.detach()), float(model.sigma.detach()), float(model.epsilon.detach())))
        if rank == 0:
            print("epoch %3d  loss %.5f  sigma %.4f  epsilon %.4f" % ((epoch,) + hist[-1]), flush=True)
    return hist


if __name__ == "__main__":
    main()
