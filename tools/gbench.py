#!/usr/bin/env python
"""Step rates of the non-fused configurations (wall clock, synchronised):
  lj4096   4096-atom LJ liquid (BASELINE config #4), NHC, forward + adjoint of an RDF loss
  gnn      CG-water SchNet + ExcludedVolume prior (BASELINE configs #3/#5 shape), Diamond lattice
Usage: python tools/gbench.py [lj4096] [gnn64] [gnn512] [--steps 10]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(integ, system, obs, nsteps, dt, method="NH_verlet", reps=3):
    from mdgrad_amd.sovlers import odeint_adjoint
    dev = system.device
    t = torch.Tensor([dt * i for i in range(nsteps + 1)]).to(dev)
    out, samples = None, []
    if os.environ.get("GBENCH_NOGC"):
        import gc
        gc.collect(); gc.freeze(); gc.disable()
    for rep in range(reps + 1):
        y0 = tuple(integ.get_inital_states(wrap=True))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        traj = odeint_adjoint(integ, y0, t, method=method)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        _, _, g = obs(traj[1][::5])
        loss = (g - 1).pow(2).mean()
        loss.backward()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if not bool(torch.isfinite(traj[1]).all()):
            raise SystemExit("gbench: non-finite trajectory -- timing would be meaningless")
        if os.environ.get("GBENCH_VERBOSE"):
            print("   rep %d fwd %.4f bwd %.4f" % (rep, t1 - t0, t2 - t1), flush=True)
        if rep > 0:
            samples.append((t1 - t0, t2 - t1))
    samples.sort(key=lambda x: x[0] + x[1])
    return samples[len(samples) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="*", default=["lj4096", "gnn64", "gnn512"])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--replicas", type=int, default=1, help="stack R replicas (System.replicate) in one trajectory")
    ap.add_argument("--table-nodes", type=int, default=0, help="mlp108: nodes of the tabulated pair energy (0 = default)")
    ap.add_argument("--bf16", action="store_true", help="gnn*: bf16 MFMA operands in the filter network (both sweeps)")
    ap.add_argument("--bf16-rows", action="store_true", help="gnn*: --bf16 and bf16 mirrors of the gathered node rows (SchNet.node_rows_bf16)")
    ap.add_argument("--freq", type=int, default=1, help="lj4096: topology_update_freq (> 1: stale lists, mdg_traj_*_large_stale)")
    ap.add_argument("--generic", action="store_true", help="lj4096 with --freq > 1: the generic path (integrator.fused_stale = False)")
    args = ap.parse_args()
    args.bf16 = args.bf16 or args.bf16_rows
    from mdgrad_amd import potentials as P, units
    from mdgrad_amd.interface import PairPotentials, GNNPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    from mdgrad_amd.nn import get_model
    from mdgrad_amd.observable import rdf
    from mdgrad_amd.system import System, Diamond, Atoms
    dev = "cuda:0"
    rng = np.random.default_rng(0)
    for w in args.which:
        if w in ("mlp4096", "mlp4096_generic"):
            # 4 096-atom liquid (BASELINE config #4 geometry) with Stack(pairMLP + LJ prior): tabulated large-N
            # fused kernels vs. the generic path (module evaluated per pair, analytic adjoint)
            n = 16
            L = (n ** 3 / 0.845) ** (1 / 3)
            g_ = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).reshape(-1, 3) * (L / n)
            pos = np.mod(g_ + rng.uniform(-0.05, 0.05, g_.shape) * (L / n), L)
            system = System(Atoms(positions=pos, cell=[L, L, L], numbers=np.ones(len(pos))), device=dev)
            system.set_velocities(rng.normal(0, 1.0, pos.shape))
            torch.manual_seed(0)
            mlp = P.pairMLP(n_gauss=25, r_start=0.0, r_end=2.5, n_layers=3, n_width=128, nonlinear="ELU")
            with torch.no_grad():
                mlp.layers[-1].weight.mul_(0.05)
            prior = PairPotentials(system, P.LennardJones(1.0, 1.0), cutoff=2.5)
            integ = NoseHooverChain(Stack({"pairnn": PairPotentials(system, mlp, cutoff=2.5), "pair": prior}), system,
                                    T=1.0, num_chains=5, Q=50.0).to(dev)
            integ.fused_table = w == "mlp4096"
            obs = rdf(system, nbins=100, r_range=(0.75, 2.5))
            tf, tb = run(integ, system, obs, args.steps, 0.005)
        elif w in ("mlp108", "mlp108_autograd"):
            # Stack(pairMLP + LJFamily prior), the LJ-fitting set-up of scripts/fit_rdf_pair.py:355-368
            from mdgrad_amd.system import FaceCenteredCubic
            atoms = FaceCenteredCubic("H", (3, 3, 3), 1.6)
            system = System(atoms, device=dev)
            system.set_positions(np.mod(system.get_positions() + rng.normal(0, 0.03, (108, 3)), 4.8))
            system.set_velocities(rng.normal(0, 1.0, (108, 3)))
            if args.replicas > 1:
                system = system.replicate(args.replicas)
                system.set_positions(np.mod(system.get_positions() + rng.normal(0, 0.03, (len(system), 3)), 4.8))
                system.set_velocities(rng.normal(0, 1.0, (len(system), 3)))
            torch.manual_seed(0)
            mlp = P.pairMLP(n_gauss=25, r_start=0.0, r_end=2.5, n_layers=3, n_width=128, nonlinear="ELU")
            with torch.no_grad():
                mlp.layers[-1].weight.mul_(0.05)
            pnn = PairPotentials(system, mlp, cutoff=2.5)
            pnn.analytic = w == "mlp108"
            prior = PairPotentials(system, P.LJFamily(epsilon=2.0, sigma=0.9, rep_pow=6, attr_pow=3), cutoff=2.5)
            integ = NoseHooverChain(Stack({"pairnn": pnn, "pair": prior}), system, T=1.0, num_chains=5, Q=50.0).to(dev)
            if args.table_nodes:
                integ.table_nodes = args.table_nodes
            obs = rdf(system, nbins=100, r_range=(0.75, 2.4))
            tf, tb = run(integ, system, obs, args.steps, 0.005)
        elif w == "lj4096":
            n = 16
            L = (n ** 3 / 0.845) ** (1 / 3)
            g = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).reshape(-1, 3) * (L / n)
            pos = np.mod(g + rng.uniform(-0.05, 0.05, g.shape) * (L / n), L)
            system = System(Atoms(positions=pos, cell=[L, L, L], numbers=np.ones(len(pos))), device=dev)
            system.set_velocities(rng.normal(0, 1.0, pos.shape))
            if args.replicas > 1:
                system = system.replicate(args.replicas)
                system.set_positions(np.mod(system.get_positions() + rng.normal(0, 0.02, (len(system), 3)), L))
                system.set_velocities(rng.normal(0, 1.0, (len(system), 3)))
            integ = NoseHooverChain(Stack({"pair": PairPotentials(system, P.LennardJones(1.0, 1.0), cutoff=2.5)}),
                                    system, T=1.0, num_chains=5, Q=50.0, topology_update_freq=args.freq).to(dev)
            if args.generic:
                integ.fused_stale = False
            obs = rdf(system, nbins=100, r_range=(0.75, 2.5))
            tf, tb = run(integ, system, obs, args.steps, 0.005)
        else:
            size = {"gnn64": 2, "gnn512": 4, "gnn4096": 8}[w]
            a = units.get_unit_len(0.997, 18.01528, 8)
            atoms = Diamond("O", (size,) * 3, a)
            pos = np.mod(atoms.get_positions() + rng.normal(0, 0.05, (len(atoms), 3)), a * size)
            atoms.set_positions(pos)
            atoms.masses[:] = 18.01528
            system = System(atoms, device=dev)
            kT = 298.0 * units.kB
            if args.replicas > 1:
                system = system.replicate(args.replicas)
                system.set_positions(np.mod(system.get_positions() + rng.normal(0, 0.05, (len(system), 3)), a * size))
            system.set_temperature(kT, rng=rng)
            torch.manual_seed(0)
            net = get_model({"n_atom_basis": 64, "n_filters": 128, "n_gaussians": 30, "n_convolutions": 2,
                             "cutoff": 6.0})
            with torch.no_grad():   # tame the random-init network so the synthetic dynamics stay finite
                net.atomwisereadout.readout["energy"][2].weight.mul_(0.02)
            net.filter_bf16 = bool(args.bf16)
            net.node_rows_bf16 = bool(args.bf16_rows)
            gnn = GNNPotentials(system, net, cutoff=6.0)
            prior = PairPotentials(system, P.ExcludedVolume(2.6, 0.01, 12), cutoff=6.0)
            integ = NoseHooverChain(Stack({"gnn": gnn, "prior": prior}), system, T=kT, num_chains=5, Q=50.0).to(dev)
            obs = rdf(system, nbins=60, r_range=(2.0, min(6.0, 0.49 * a * size)))
            tf, tb = run(integ, system, obs, args.steps, 1.0 * units.fs)
        R = getattr(system, "n_replicas", 1)
        print("%-8s N=%5d x %d replicas  steps=%d  fwd %.4f s  bwd(adjoint+rdf) %.4f s  -> %.1f MD steps/s (fwd+adj, all replicas)" % (
            w, system.group_size, R, args.steps, tf, tb, R * args.steps / (tf + tb)), flush=True)


if __name__ == "__main__":
    main()
