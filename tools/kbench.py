#!/usr/bin/env python
"""Kernel-level timing of the fused 108-atom path (HIP events on the launch stream):
forward trajectory, adjoint sweep, rdf forward/backward, for a sweep of replica counts and
workgroup sizes.  Usage: python tools/kbench.py [--reps 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--frames", type=int, default=50)
    ap.add_argument("--configs", default="256:1024,1024:256,1024:512,4096:256,4096:448,4096:512,8192:256")
    args = ap.parse_args()
    from mdgrad_amd import ops
    from mdgrad_amd import potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    from mdgrad_amd.observable import rdf
    from mdgrad_amd.system import System
    dev = torch.device("cuda:0")
    T = args.frames
    print("%8s %6s | %9s %9s %9s %9s | %12s %12s" % ("R", "block", "fwd ms", "adj ms", "rdf_f ms", "rdf_b ms",
                                                     "fwd Msteps/s", "f+a Msteps/s"))
    for cfg in args.configs.split(","):
        R, block = [int(x) for x in cfg.split(":")]
        atoms, pos, vel = bench.make_inputs(R, 7, dev)
        system = System(atoms, device=dev)
        integ = NoseHooverChain(Stack({"pair": PairPotentials(system, P.LennardJones(1.0, 1.0), cutoff=2.5)}),
                                system, T=1.0, num_chains=5, Q=50.0).to(dev)
        obs = rdf(system, nbins=100, r_range=(0.75, 2.5))
        t = torch.Tensor([0.005 * i for i in range(T)]).to(dev)
        pv0 = torch.zeros(R, 5, device=dev)
        spec = integ.fused_spec("NH_verlet")
        spec.block = block
        theta = spec.flat_params()
        out = {}

        def fwd():
            out["y"] = ops.FusedTrajFn.apply(vel, pos, pv0, t, theta, spec)
        t_f = timeit(fwd, args.reps)
        v_t, q_t, pv_t = out["y"]
        gq = torch.randn_like(q_t) * 1e-3

        def adj():
            torch.autograd.grad(q_t, theta, gq, retain_graph=True)
        t_a = timeit(adj, args.reps)
        qd = q_t.detach().requires_grad_(True)

        def rf():
            out["g"] = obs(qd)[2]
        t_rf = timeit(rf, args.reps)
        gg = torch.randn_like(out["g"])

        def rb():
            torch.autograd.grad(out["g"], qd, gg, retain_graph=True)
        t_rb = timeit(rb, args.reps)
        steps = R * (T - 1)
        print("%8d %6d | %9.3f %9.3f %9.3f %9.3f | %12.2f %12.2f" % (
            R, block, t_f, t_a, t_rf, t_rb, steps / t_f / 1e3, steps / (t_f + t_a) / 1e3))


if __name__ == "__main__":
    main()
