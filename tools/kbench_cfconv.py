#!/usr/bin/env python
"""Timings of the fused SchNet interaction-block kernels and the node-level Dense kernel on the topology of
bench.py --workload schnet4096 (4096 CG-water beads x R stacked replicas, cutoff 6, A64 / F128 / G30):
    python tools/kbench_cfconv.py [--replicas 8] [--reps 20] [--bf16 | --rows16]
HIP events on the launch stream; also the vehicle for rocprofv3 --pmc passes (tools/pmc_cfconv.sh)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replicas", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--rows16", action="store_true", help="--bf16 with bf16 mirrors of the gathered node rows (mdg_cfconv_*_rows16)")
    ap.add_argument("--cold", action="store_true", help="each timed launch after a 1 GiB fill (L2 and the 256 MB Infinity Cache hold "
                    "none of its operands): what a sweep costs inside a pass, between kernels that touch other data")
    args = ap.parse_args()
    args.bf16 = args.bf16 or args.rows16
    from mdgrad_amd import ops, units, _lib
    from mdgrad_amd.system import System, Diamond
    dev = "cuda:0"
    rng = np.random.default_rng(0)
    a = units.get_unit_len(0.997, 18.01528, 8)
    atoms = Diamond("O", (8,) * 3, a)
    base = System(atoms, device=dev)
    system = base.replicate(args.replicas) if args.replicas > 1 else base
    L = a * 8
    pos = np.mod(system.get_positions() + rng.normal(0, 0.05, (len(system), 3)), L).astype(np.float32)
    x = torch.from_numpy(pos).to(dev)
    cs = _lib.make_cell(np.array([L, L, L], dtype=np.float32))
    ell = ops.build_ell(x, cs, 6.0, group=len(atoms))
    topo = ops.GraphTopo(ell)
    N, E, G, F, A = topo.n_atoms, topo.n_edges, 30, 128, 64
    torch.manual_seed(0)
    mu = torch.linspace(0, 6.0, G, device=dev)
    coef = torch.full((G,), -0.5 / float(mu[1] - mu[0]) ** 2, device=dev)
    net = (mu, coef, torch.randn(G, G, device=dev) / G ** 0.5, torch.randn(G, device=dev) * 0.1,
           torch.randn(F, G, device=dev) / G ** 0.5, torch.randn(F, device=dev) * 0.1)
    fn = ops.FilterNet(*net, bf16=args.bf16, rows16=args.rows16)
    w = torch.randn(N, 3, device=dev)
    d, uhat, dd, ddel = ops.edge_geom(x, topo, w)
    h, hd, mb, mdb = [torch.randn(N, F, device=dev) for _ in range(4)]
    hf, hdf = h, hd                               # (f32 rows for the Dense case below)
    if args.rows16:
        assert fn.rows16
        h, hd, mb, mdb = [ops.rows_to_bf16(v) for v in (h, hd, mb, mdb)]
    d_b, dd_b = torch.zeros(E, device=dev), torch.zeros(E, device=dev)
    r, rd = torch.randn(N, A, device=dev), torch.randn(N, A, device=dev)
    Wn, bn = torch.randn(F, A, device=dev) / 8, torch.randn(F, device=dev)
    U1, c1 = torch.randn(A, F, device=dev) / 11, torch.randn(A, device=dev)
    tiles = int(((ell.cnt + 15) // 16).sum())
    print("N=%d E=%d directed 16-slot tiles=%d (%.1f%% rows used) edge tiles=%d" % (
        N, E, tiles, 100.0 * 2 * E / (16.0 * tiles), (E + 15) // 16))
    cases = [
        ("cfconv_fwd primal", lambda: ops.cfconv_fwd(fn, d, None, h, None, topo), tiles * 80),
        ("cfconv_fwd primal+tangent", lambda: ops.cfconv_fwd(fn, d, dd, h, hd, topo), tiles * 160),
        ("cfconv_bwd plain", lambda: ops.cfconv_bwd(fn, d, None, topo, h, None, None, mdb, None, dd_b), ((E + 15) // 16) * 96),
        ("cfconv_bwd dual", lambda: ops.cfconv_bwd(fn, d, dd, topo, h, hd, mb, mdb, d_b, dd_b), ((E + 15) // 16) * 192),
        ("cfconv_bwd dual+theta", lambda: ops.cfconv_bwd(fn, d, dd, topo, h, hd, mb, mdb, d_b, dd_b, True), ((E + 15) // 16) * 352),
        ("cfconv_fwd primal+tangent (no hd)", lambda: ops.cfconv_fwd(fn, d, dd, h, None, topo), tiles * 160),
        ("cfconv_bwd dual (no hd)", lambda: ops.cfconv_bwd(fn, d, dd, topo, h, None, mb, mdb, d_b, dd_b), ((E + 15) // 16) * 192),
        ("cfconv_bwd dual+theta (no hd)", lambda: ops.cfconv_bwd(fn, d, dd, topo, h, None, mb, mdb, d_b, dd_b, True), ((E + 15) // 16) * 352),
        ("dense A->F dual", lambda: ops.dense(Wn, r, bias=bn, x1=rd), 0),
        ("dense F->A ssp dual", lambda: ops.dense(U1, hf, bias=c1, act=True, x1=hdf, want_sig=True), 0),
        ("rows_to_bf16 [N,F]", lambda: ops.rows_to_bf16(hf), 0),
        ("edge_geom (tangent)", lambda: ops.edge_geom(x, topo, w), 0),
        ("edge_geom_bwd", lambda: ops.edge_geom_bwd(d_b, dd_b, d, dd, uhat, ddel, topo), 0),
        ("nbr build (cell, grouped)", lambda: ops.build_ell(x, cs, 6.0, group=len(atoms), max_nbr=ell.max_nbr), 0),
    ]
    if args.bf16:
        st_s, st_sd = ops.cfconv_filter_stash(fn, d, dd, topo)
        for tag, a, b in (("primal", ops.cfconv_fwd(fn, d, None, h, None, topo)[0], ops.cfconv_fwd_stashed(fn, st_s, None, d, h, None, topo)[0]),
                          ("tangent m", ops.cfconv_fwd(fn, d, dd, h, hd, topo)[0], ops.cfconv_fwd_stashed(fn, st_s, st_sd, d, h, hd, topo)[0]),
                          ("tangent md", ops.cfconv_fwd(fn, d, dd, h, hd, topo)[1], ops.cfconv_fwd_stashed(fn, st_s, st_sd, d, h, hd, topo)[1]),
                          ("tangent md (no hd)", ops.cfconv_fwd(fn, d, dd, h, None, topo)[1], ops.cfconv_fwd_stashed(fn, st_s, st_sd, d, h, None, topo)[1])):
            print("stashed == recomputed, %-18s: %s (max |d| %.3e)" % (tag, bool(torch.equal(a, b)), float((a - b).abs().max())))
        cases += [
            ("filter_stash (s)", lambda: ops.cfconv_filter_stash(fn, d, None, topo), 0),
            ("filter_stash (s, sd)", lambda: ops.cfconv_filter_stash(fn, d, dd, topo), 0),
            ("cfconv_fwd STASHED primal", lambda: ops.cfconv_fwd_stashed(fn, st_s, None, None, h, None, topo), 0),
            ("cfconv_fwd STASHED primal+tangent", lambda: ops.cfconv_fwd_stashed(fn, st_s, st_sd, None, h, hd, topo), 0),
            ("cfconv_fwd STASHED tangent (no hd)", lambda: ops.cfconv_fwd_stashed(fn, st_s, st_sd, None, h, None, topo), 0),
        ]
    junk = torch.empty(1 << 28, device=dev) if args.cold else None
    for name, fnc, mfma in cases:
        fnc()
        if args.cold:
            tot = 0.0
            for _ in range(args.reps):
                junk.fill_(1.0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fnc()
                e1.record()
                torch.cuda.synchronize()
                tot += e0.elapsed_time(e1)
            ms = tot / args.reps
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fnc()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.reps
        extra = ""
        if mfma and not args.bf16:
            tf = mfma * 2048.0 / (ms * 1e-3) / 1e12
            extra = "  %6.1f TFLOP/s executed MFMA = %.0f%% of 157.3" % (tf, 100 * tf / 157.3)
        print("%-34s %8.1f us%s" % (name, ms * 1e3, extra), flush=True)


if __name__ == "__main__":
    main()
