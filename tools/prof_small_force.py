#!/usr/bin/env python
"""Which launches does ONE force / force-vjp evaluation of the config-#3 integrator (192-atom water, SchNet + prior) make?
torch.profiler over an eager evaluation: aten ops that launch kernels (copies, fills, elementwise) next to the library's own
kernels -- the small launches a HIP-graph replay of the step still pays ~4-5 us each for.   gpurun -- python tools/prof_small_force.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = "cuda:0"
    g, sd, prm, system, net, gnn, integ = bench.build_water192(dev)
    v, q, pv = (x.clone() for x in integ.get_inital_states(wrap=True))
    model = integ.model
    w = torch.randn_like(q)
    for _ in range(3):
        integ.update_topology(q)
        model.force(q)
        model.force_vjp(q, w, want_theta=True)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    for name, fn in (("update_topology + force", lambda: (integ.update_topology(q), model.force(q))),
                     ("update_topology + force_vjp (theta)", lambda: (integ.update_topology(q), model.force_vjp(q, w, want_theta=True)))):
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            fn()
            torch.cuda.synchronize()
        print("==", name)
        rows = [e for e in prof.key_averages() if e.device_time_total > 0 or e.key.startswith("aten::")]
        rows.sort(key=lambda e: -e.count)
        for e in rows[:40]:
            print("%-70s calls %3d  device %8.1f us" % (e.key[:70], e.count, e.device_time_total))


if __name__ == "__main__":
    main()
