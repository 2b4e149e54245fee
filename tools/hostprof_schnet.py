#!/usr/bin/env python
"""Where the HOST spends its time in one eager pass of bench.py's stacked SchNet workload (8 x 4096 beads: more pairs than
graphs.MAX_EDGES, so every launch is issued from Python):  python tools/hostprof_schnet.py [--bf16-rows | --f32] [--passes 3]
Prints the pass time with the GPU running asynchronously, the host-only time of the same pass (kernel launches are
asynchronous: the time until the last launch is issued), and the cProfile top of the host side."""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bf16-rows", action="store_true")
    ap.add_argument("--f32", action="store_true")
    ap.add_argument("--opt", type=int, default=0, help="instead of the profile: N passes WITH an Adam step each, time per pass")
    ap.add_argument("--passes", type=int, default=3)
    args = ap.parse_args()
    import bench
    from mdgrad_amd import units
    from mdgrad_amd.observable import rdf
    from mdgrad_amd.sovlers import odeint_adjoint
    dev = torch.device("cuda:0")
    wl = bench.build_schnet_workload(dev, 8, not args.f32, 2000, rows16=args.bf16_rows)
    integ, system = wl["integ"], wl["system"]
    obs = rdf(system, nbins=60, r_range=(2.0, 6.0))
    target = torch.ones(60, device=dev)
    t = torch.Tensor([units.fs * i for i in range(11)]).to(dev)
    params = list(integ.parameters())

    y0_dev = tuple(x.clone() for x in integ.get_inital_states(wrap=True))

    def one():
        for p in params:
            p.grad = None
        y0 = tuple(x.clone() for x in y0_dev)
        v_t, q_t, pv_t = odeint_adjoint(integ, y0, t, method="NH_verlet")
        loss = (obs(q_t[::5])[2] - target).pow(2).mean()
        loss.backward()
        return loss

    for _ in range(2):
        one()
    torch.cuda.synchronize()
    if args.opt:
        import gc
        gcs = []

        def on_gc(phase, info, _t=[0.0]):                      # how long every collection of the cyclic collector takes
            if phase == "start":
                _t[0] = time.perf_counter()
            else:
                gcs.append((info["generation"], (time.perf_counter() - _t[0]) * 1e3, info["collected"]))
        gc.callbacks.append(on_gc)
        if os.environ.get("MDG_NO_GC") == "1":
            gc.collect()
            gc.disable()
        opt = torch.optim.Adam(params, lr=1e-5)
        vl = (wl["gnn"]._static or {}).get("verlet")
        for k in range(args.opt):
            b0 = vl.builds() if vl is not None else 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            opt.zero_grad(set_to_none=True)
            one()
            t1 = time.perf_counter()
            opt.step()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print("pass %2d: fwd + adjoint %.2f ms, + Adam step %.2f ms, searches %d, capacity %s" % (
                k, (t1 - t0) * 1e3, (t2 - t0) * 1e3, (vl.builds() - b0) if vl is not None else -1,
                {kk: vv for kk, vv in (wl["gnn"]._static or {}).items() if kk in ("max_nbr", "capacity", "version")}), flush=True)
            if gcs:
                print("         collections during this pass (generation, ms, objects freed): %s" % [(g, round(ms, 2), c) for g, ms, c in gcs], flush=True)
                del gcs[:]
        return
    for _ in range(args.passes):
        t0 = time.perf_counter()
        one()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("pass: host issued everything after %.2f ms, GPU done after %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
    # the adjoint sweep runs inside loss.backward(): on the calling thread here, so that the profile sees it
    with torch.autograd.set_multithreading_enabled(False):
        one()
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(args.passes):
            one()
        pr.disable()
    torch.cuda.synchronize()
    for key, n in (("tottime", 60), ("cumulative", 70)):
        out = io.StringIO()
        pstats.Stats(pr, stream=out).sort_stats(key).print_stats(n)
        print(out.getvalue())


if __name__ == "__main__":
    main()
