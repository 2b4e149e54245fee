#!/usr/bin/env python
"""Rate of the wave-per-replica kernels with a masked term (two-species index_tuple) against the unmasked single-form
ring: 108-atom LJ, R replicas x 49 steps forward + adjoint (no observable).  python tools/kbench_ring_mask.py [--replicas R]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replicas", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    from mdgrad_amd import ops, potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    from mdgrad_amd.system import System, FaceCenteredCubic
    dev = "cuda:0"
    R, T = args.replicas, 50
    rng = np.random.default_rng(0)
    atoms = FaceCenteredCubic(symbol="H", size=(3, 3, 3), latticeconstant=1.6)
    system = System(atoms, device=dev)
    lat = atoms.get_positions()
    pos = torch.from_numpy(np.mod(lat[None] + rng.uniform(-0.05, 0.05, (R,) + lat.shape), 4.8).astype(np.float32)).to(dev)
    vel = torch.from_numpy(rng.normal(0, 1.0, pos.shape).astype(np.float32)).to(dev)
    t = torch.Tensor([0.005 * i for i in range(T)]).to(dev)
    A_, B_ = list(range(0, 108, 2)), list(range(1, 108, 2))
    mix = [dict(index_tuple=(A_, A_)), dict(index_tuple=(B_, B_)), dict(index_tuple=(A_, B_))]
    for name, kws, block in (("unmasked LJ 12-6", [{}], 0), ("two species, index_tuple (A, B)", [dict(index_tuple=(A_, B_))], 0),
                             ("mixture A-A + B-B + A-B, ring (one shared sweep)", mix, 0),
                             ("mixture A-A + B-B + A-B, workgroup kernels", mix, 128)):
        mdls = [P.LennardJones(1.0 - 0.05 * k, 1.0 + 0.1 * k) for k in range(len(kws))]
        integ = NoseHooverChain(Stack({"t%d" % k: PairPotentials(system, m, cutoff=2.5, **kw) for k, (m, kw) in enumerate(zip(mdls, kws))}),
                                system, T=1.0, num_chains=5, Q=50.0).to(dev)
        integ.fuse_observables = False
        spec = integ.fused_spec("NH_verlet")
        spec.block = block
        pv0 = torch.zeros(R, 5, device=dev)

        def one():
            v0, q0 = vel.clone().requires_grad_(True), pos.clone().requires_grad_(True)
            v_t, q_t, pv_t = ops.FusedTrajFn.apply(v0, q0, pv0, t, spec.flat_params(), spec)
            (q_t[:, -1].pow(2).mean() + v_t[:, -1].pow(2).mean()).backward()
        one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            one()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / args.reps
        print("%-52s %8.2f ms per pass  %6.2f M MD steps/s (fwd + adjoint, %d replicas)" % (name, el * 1e3, R * (T - 1) / el / 1e6, R))


if __name__ == "__main__":
    main()
