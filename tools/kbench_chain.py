#!/usr/bin/env python
"""Launch time of mdg_row_chain (csrc/rowchain.hip) against the same layers as separate mdg_dense launches.
    python tools/kbench_chain.py [--rows 4096] [--a 64] [--f 128]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps=200):
    for _ in range(10):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps // 20):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (reps // 20 * 20)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4096)
    ap.add_argument("--a", type=int, default=64)
    ap.add_argument("--f", type=int, default=128)
    ap.add_argument("--flag", type=int, default=0, help="0: f32 matrix instruction, 2: MDG_CHAIN_X3, 4: MDG_CHAIN_X6")
    args = ap.parse_args()
    from mdgrad_amd import ops, _lib
    dev = torch.device("cuda:0")
    N, A, F = args.rows, args.a, args.f
    rn = lambda *s: torch.randn(*s, device=dev)
    U1, U2, L1, Wn = rn(A, F) / F ** 0.5, rn(A, A) / A ** 0.5, rn(A // 2, A) / A ** 0.5, rn(F, A) / A ** 0.5
    c1, c2, l1, bn, L2 = rn(A), rn(A), rn(A // 2), rn(F), rn(1, A // 2)
    m, md, r, rd = rn(N, F), rn(N, F), rn(N, A), rn(N, A)
    keep = {}

    def chain(kind, dual):
        ch = ops.RowChain(N, dual, dev, args.flag)
        a = ch.stage(U1, bias=c1, act=True, in0=m, in1=md if dual else None, want_sig=True)
        if kind >= 2:
            ch.stage(U2, bias=c2, res0=r, res1=rd if dual else None)
        if kind == 3:
            ch.stage(Wn, bias=bn)
        if kind == 6:
            ch.stage(L1, bias=l1, act=True, mode=_lib.CHAIN_HEAD, aux0=L2, want_sig=True, want_pre=(False, True))
            ch.stage(L1, trans=True)
            if dual:
                ch.stage(U2, trans=True, mode=_lib.CHAIN_SSP_BWD, aux0=a.sig, aux1=a.out1)
            else:
                ch.stage(U2, trans=True, mode=_lib.CHAIN_MUL, aux0=a.sig)
            ch.stage(U1, trans=True)
        ch.run()
        keep[(kind, dual)] = ch

    def layers(kind, dual):
        t, su, td = ops.dense(U1, m, bias=c1, act=True, x1=md if dual else None, want_sig=True)
        if kind >= 2:
            rr, _, rrd = ops.dense(U2, t, bias=c2, res=r, x1=td, res1=rd if dual else None)
        if kind == 3:
            ops.dense(Wn, rr, bias=bn, x1=rrd)

    print("rows %d, A %d, F %d" % (N, A, F))
    for dual in (False, True):
        for kind in (1, 2, 3, 6):
            tc = timed(lambda: chain(kind, dual))
            tl = timed(lambda: layers(kind, dual)) if kind <= 3 else float("nan")
            print("  %s %d stage(s): chain %6.1f us   separate dense launches %6.1f us" % ("dual  " if dual else "single", kind, tc, tl))


if __name__ == "__main__":
    main()
