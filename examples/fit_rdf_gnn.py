#!/usr/bin/env python
"""Fit a SchNet coarse-grained water potential to a target RDF -- the training loop of the reference's
demo/fit_rdf_gnn.py:215-513 without plotting: CG-water Diamond box, SchNet + ExcludedVolume prior in a
Stack, NoseHooverChain, `Simulations` epochs of `tau` steps continuing from the last frame, temperature
annealing through `update_T`, RDF on every 20th frame, JS divergence + volume-weighted deviation
(`compute_D`) loss, Adam + ReduceLROnPlateau.

BASELINE config #5 ("8 replica trajectories x SchNet, bf16 cfconv MFMA, full fwd+adjoint, 8 x MI355X"): the
`--replicas` independent trajectories (replica k is seeded with k, like the reference's `sim_list` loop,
demo/fit_rdf_gnn.py:386-399) are sharded over the ranks with mdgrad_amd.dist -- one process per GPU, the
replicas of a rank stacked in one state (System.replicate) -- forward + adjoint need no communication, ONE
all-reduce of the parameter gradient per epoch precedes identical optimizer steps on every rank.  `--bf16` runs
the filter network of the fused interaction block on bf16 MFMA operands.

    python examples/fit_rdf_gnn.py --size 4 --replicas 8 --epochs 20
    python examples/fit_rdf_gnn.py --gpus 8 --size 8 --replicas 8 --bf16     # spawns its own 8 ranks (RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29512 \
        examples/fit_rdf_gnn.py --size 8 --replicas 8 --bf16
    python examples/fit_rdf_gnn.py --target my_rdf.csv     # (r, g) columns instead of the synthetic target
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def synthetic_target(r):
    """A water-like g(r): excluded core, first peak near 2.8 A, damped oscillation."""
    core = 1.0 / (1.0 + np.exp(-(r - 2.55) / 0.08))
    return core * (1.0 + 1.6 * np.exp(-((r - 2.85) / 0.28) ** 2) + 0.25 * np.exp(-((r - 4.6) / 0.6) ** 2)
                   - 0.2 * np.exp(-((r - 3.5) / 0.4) ** 2))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4, help="Diamond cells per side (8 size^3 beads)")
    ap.add_argument("--replicas", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--tau", type=int, default=40, help="MD steps per epoch (opt_freq)")
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--start-T", type=float, default=350.0)
    ap.add_argument("--T", type=float, default=298.0)
    ap.add_argument("--anneal-rate", type=float, default=5.0)
    ap.add_argument("--target", default=None)
    ap.add_argument("--filters", type=int, default=128, help="n_filters of the SchNet (config #5: 128)")
    ap.add_argument("--bf16", action="store_true", help="bf16 MFMA operands in the cfconv filter network")
    ap.add_argument("--gpus", type=int, default=1, help="ranks to spawn (one per GPU) when started without a launcher")
    args = ap.parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from mdgrad_amd import dist as mdist
        sys.exit(mdist.self_launch(os.path.abspath(__file__), sys.argv[1:] if argv is None else list(argv), args.gpus))
    from mdgrad_amd import fit, potentials as P, units
    from mdgrad_amd.interface import GNNPotentials, PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain, Simulations
    from mdgrad_amd.nn import get_model
    from mdgrad_amd.observable import rdf
    from mdgrad_amd.system import System, Diamond
    from mdgrad_amd import dist as mdist
    rank, world, dev = mdist.init()
    lo, hi = mdist.shard_range(args.replicas, rank, world)      # this rank's replicas
    if hi == lo:
        raise SystemExit("fit_rdf_gnn: more ranks than replicas")
    a = fit.get_unit_len(0.997, 18.01528, 8)
    atoms = Diamond("O", (args.size,) * 3, a)
    atoms.masses[:] = 18.01528
    system = System(atoms, device=dev)
    if hi - lo > 1:
        system = system.replicate(hi - lo)
    L = a * args.size
    lat = atoms.get_positions()
    pos, vel = [], []
    for k in range(lo, hi):                                     # replica k: its own seed, whatever the sharding
        rk = np.random.default_rng(k)
        pos.append(np.mod(lat + rk.normal(0, 0.05, lat.shape), L))
        vel.append(rk.normal(0, np.sqrt(args.start_T * units.kB / 18.01528), lat.shape))
    system.set_positions(np.concatenate(pos))
    system.set_velocities(np.concatenate(vel))

    cutoff, nbins = 6.0, 60
    r_range = (2.0, min(cutoff, 0.49 * L))
    data = (np.loadtxt(args.target, delimiter=",") if args.target else
            np.stack([np.linspace(1.5, 8.0, 400), synthetic_target(np.linspace(1.5, 8.0, 400))]))
    bins, g_target = fit.get_exp_rdf(data, nbins, r_range, dev)

    torch.manual_seed(0)
    net = get_model({"n_atom_basis": 64, "n_filters": args.filters, "n_gaussians": 30, "n_convolutions": 2,
                     "cutoff": cutoff})
    net.filter_bf16 = bool(args.bf16)
    with torch.no_grad():                                       # start from a gentle correction to the prior
        net.atomwisereadout.readout["energy"][2].weight.mul_(0.02)
    prior = P.ExcludedVolume(2.6, 0.01, 12)
    model = Stack({"gnn": GNNPotentials(system, net, cutoff=cutoff), "prior": PairPotentials(system, prior, cutoff=cutoff)})
    integ = NoseHooverChain(model, system, T=args.start_T * units.kB, num_chains=5, Q=50.0).to(dev)
    sim = Simulations(system, integ)
    obs = rdf(system, nbins=nbins, r_range=r_range)
    opt = torch.optim.Adam(net.parameters(), lr=args.lr)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, "min", min_lr=0.9e-7, factor=0.5, patience=25, threshold=1e-5)
    rho = system.group_size / (L ** 3)
    rrange = torch.linspace(float(bins[0]), float(bins[-1]), nbins, device=dev)
    hist = []
    for i in range(args.epochs):
        new_T = fit.get_temp(args.start_T, args.T, args.epochs, i, args.anneal_rate)
        sim.integrator.update_T(new_T * units.kB)
        v_t, q_t, pv_t = sim.simulate(steps=args.tau, frequency=args.tau, dt=1.0 * units.fs)
        if torch.isnan(q_t).any():
            raise RuntimeError("trajectory diverged")
        g = obs(q_t[::20])[2]
        loss_js = fit.JS_rdf(g_target, g)
        loss = fit.compute_D(g - g_target, rho, rrange)
        loss.backward()
        mdist.all_reduce_grads(net.parameters(), average=True)   # the one collective per outer step
        opt.step()
        opt.zero_grad()
        gl = mdist.sum_over_ranks(float(loss.detach()), dev) / world
        gjs = mdist.sum_over_ranks(float(loss_js.detach()), dev) / world
        sched.step(gl)
        hist.append((gl, gjs, new_T))
        if rank == 0:
            print("epoch %3d | T %.1f K | loss %.5f | JS %.5f" % (i, new_T, gl, gjs), flush=True)
    # every rank applied the same updates: the parameters must agree bit for bit
    checksum = float(sum(p.detach().double().sum() for p in net.parameters()))
    print("PARAM_CHECKSUM rank %d of %d replicas [%d,%d) %.12e" % (rank, world, lo, hi, checksum), flush=True)
    import torch.distributed as tdist
    if tdist.is_available() and tdist.is_initialized():
        mdist.barrier()
        tdist.destroy_process_group()
    return hist


if __name__ == "__main__":
    main()
