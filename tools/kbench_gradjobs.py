#!/usr/bin/env python
"""Launch time of mdg_grad_jobs (csrc/gradjobs.hip) by job kind, at the sizes of one SchNet adjoint evaluation.
    python tools/kbench_gradjobs.py [--rows 4096]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kbench_chain import timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4096)
    args = ap.parse_args()
    from mdgrad_amd import ops
    dev = torch.device("cuda:0")
    N, A, F = args.rows, 64, 128
    rn = lambda *s: torch.randn(*s, device=dev)
    xa, xb, xc, xd = rn(N, A), rn(N, A), rn(N, A), rn(N, A)
    fa, fb, fc, fd = rn(N, F), rn(N, F), rn(N, F), rn(N, F)
    ya, yb = rn(N, A // 2), rn(N, A // 2)
    sizes = [A * A, A * F, F * A, A * A // 2, A, F, A // 2]
    params = [torch.nn.Parameter(torch.zeros(s, device=dev)) for s in sizes * 3]
    acc = ops.ThetaAccum(params)
    off = [acc.off[id(p)] for p in params]

    def run(kinds):
        jobs = ops.GradJobs()
        if "atb" in kinds:
            for i in range(2):
                jobs.atb(off[0 + 7 * i], xa, xb, xc, xd)          # U2: [A, A]
                jobs.atb(off[1 + 7 * i], xa, fa, xc, fb)          # U1: [A, F]
                jobs.atb(off[2 + 7 * i], fa, xa, fb, xb)          # Wn: [F, A]
            jobs.atb(off[3], ya, xa, yb, xb)                      # L1: [A/2, A]
        if "colsum" in kinds:
            for i in range(2):
                jobs.colsum(off[4 + 7 * i], xa)
                jobs.colsum(off[4 + 7 * i + 7], xb)
                jobs.colsum(off[5 + 7 * i], fa, fb, fc, fd)
                jobs.colsum(off[5 + 7 * i + 7], fa)
            jobs.colsum(off[6], ya)
            jobs.colsum(off[6 + 7], yb)
        jobs.run(acc, alpha=-1.0, accumulate=True)

    print("rows %d" % N)
    for kinds in (("atb",), ("colsum",), ("atb", "colsum")):
        print("  %-16s %6.1f us per call (partial + reduce launches)" % ("+".join(kinds), timed(lambda: run(kinds))))


if __name__ == "__main__":
    main()
