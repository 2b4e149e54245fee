#!/usr/bin/env python
"""Headline benchmark: MD steps/sec (forward + adjoint) for the 108-atom LJ system of
BASELINE.json configs[1] on N MI355X GPUs.

One "step" of this script = one pass of the hot path over one batch of synthetic input:
R independent replicas x (T-1) NH-Verlet steps forward (one fused launch), the soft-histogram
RDF loss and its gradient, and the full adjoint sweep (one fused launch), plus -- for N > 1 --
the single all-reduce of the parameter gradient.  value = R*(T-1)*N_gpus*K / time.
Inputs are generated once and are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29511 bench.py --gpus 8 --steps 10 --warmup 2
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def make_inputs(R, seed, dev):
    from mdgrad_amd.system import FaceCenteredCubic
    rng = np.random.default_rng(seed)
    atoms = FaceCenteredCubic(symbol="H", size=(3, 3, 3), latticeconstant=1.6)
    lat = atoms.get_positions()
    pos = np.mod(lat[None] + rng.uniform(-0.05, 0.05, (R,) + lat.shape), 4.8).astype(np.float32)
    vel = rng.normal(0.0, np.sqrt(1.0 / 1.008), pos.shape).astype(np.float32)
    return atoms, torch.from_numpy(pos).to(dev), torch.from_numpy(vel).to(dev)


def cpu_baseline(frames, dt, budget_s=12.0):
    """The CPU oracle (port of the reference algorithm, oracle/) on this host: the same 108-atom
    workload, one replica at a time, forward + rdf loss + adjoint; bounded to ~budget_s."""
    import oracle as O
    # 108-atom tensors are far too small for one thread per core on a 100+-core host
    nthreads = min(8, os.cpu_count() or 1)
    torch.set_num_threads(nthreads)
    _, pos, vel = make_inputs(1, 123, "cpu")
    cell = torch.tensor([4.8] * 3)
    t = torch.Tensor([dt * i for i in range(frames)])

    def one():
        term = O.PairTerm("lj", torch.tensor([1.0, 1.0]), 2.5, cell, p=12, q=6, c=1)
        eom = O.NHCOracle(O.ModelOracle([term]), torch.full((108,), 1.008), 1.0, 50.0, 5)
        traj = O.odeint_oracle(eom, (vel[0], pos[0], torch.zeros(5)), t)
        leaves = [x.clone().requires_grad_(True) for x in traj]
        _, _, g = O.rdf_oracle(leaves[1], cell, 100, (0.75, 2.5))
        (g - 1).pow(2).mean().backward()
        O.adjoint_oracle(eom, traj, [x.grad for x in leaves], t)

    one()                                   # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        one()
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 200:
            break
    return {"value": n * (frames - 1) / el, "unit": "MD steps/s", "cores": nthreads, "kind": "port",
            "sample": "%d trajectories x %d steps (fwd + rdf loss + adjoint), 108-atom LJ, oracle/ on %d "
                      "torch threads, %.1f s" % (n, frames - 1, nthreads, el)}


def schnet_workload(args, rank, world, dev, mdist):
    """Second headline (north_star: 4 096-bead SchNet water): CG-water Diamond 8^3 box, SchNet
    A64/F128/G30/2 conv + ExcludedVolume prior, NoseHooverChain, R stacked replicas per GPU,
    forward + RDF loss + analytic adjoint + grad all-reduce + Adam.  `--replicas` = replicas per GPU
    (default 4 for this workload), `--frames` = saved frames."""
    from mdgrad_amd import ops, potentials as P, units
    from mdgrad_amd.interface import GNNPotentials, PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    from mdgrad_amd.nn import get_model
    from mdgrad_amd.observable import rdf
    from mdgrad_amd.sovlers import odeint_adjoint
    from mdgrad_amd.system import System, Diamond
    R, T = args.replicas, args.frames
    rng = np.random.default_rng(2000 + rank)
    a = units.get_unit_len(0.997, 18.01528, 8)
    size = 8
    atoms = Diamond("O", (size,) * 3, a)
    atoms.masses[:] = 18.01528
    base = System(atoms, device=dev)
    system = base.replicate(R) if R > 1 else base
    L = a * size
    system.set_positions(np.mod(system.get_positions() + rng.normal(0, 0.05, (len(system), 3)), L))
    kT = 298.0 * units.kB
    system.set_temperature(kT, rng=rng)
    torch.manual_seed(0)
    net = get_model({"n_atom_basis": 64, "n_filters": 128, "n_gaussians": 30, "n_convolutions": 2, "cutoff": 6.0})
    with torch.no_grad():        # random-init SchNet forces are O(100 eV/A): scale the readout so the synthetic
        net.atomwisereadout.readout["energy"][2].weight.mul_(0.02)   # dynamics stay stable (checked below)
    integ = NoseHooverChain(Stack({"gnn": GNNPotentials(system, net, cutoff=6.0),
                                   "prior": PairPotentials(system, P.ExcludedVolume(2.6, 0.01, 12), cutoff=6.0)}),
                            system, T=kT, num_chains=5, Q=50.0).to(dev)
    obs = rdf(system, nbins=60, r_range=(2.0, 6.0))
    target = torch.ones(60, device=dev)
    t = torch.Tensor([units.fs * i for i in range(T)]).to(dev)
    params = list(integ.parameters())
    opt = torch.optim.Adam(params, lr=1e-5)

    def step():
        opt.zero_grad(set_to_none=True)
        y0 = tuple(integ.get_inital_states(wrap=True))
        v_t, q_t, pv_t = odeint_adjoint(integ, y0, t, method="NH_verlet")
        loss = (obs(q_t[::5])[2] - target).pow(2).mean()
        loss.backward()
        mdist.all_reduce_grads(params)
        opt.step()
        return loss, q_t

    for _ in range(args.warmup):
        step()
    mdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, q_last = step()
    torch.cuda.synchronize()
    mdist.barrier()
    el = mdist.max_over_ranks(time.perf_counter() - t0, dev)
    if not (bool(torch.isfinite(q_last).all()) and all(bool(torch.isfinite(p).all()) for p in params)):
        raise SystemExit("bench: non-finite trajectory or parameters -- the measurement would be invalid")
    N = base.get_number_of_atoms()
    md_steps = R * (T - 1) * world * args.steps
    out = {"metric": "MD steps/sec (fwd+adjoint), 4096-bead SchNet CG water NHC", "value": md_steps / el,
           "unit": "MD steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "CG water Diamond 8^3 (%d beads), SchNet A64 F128 G30 2 conv + ExcludedVolume prior, "
                                  "cutoff 6, NoseHooverChain(Q=50, 5 chains), %d steps fwd + RDF(60 bins) loss + analytic "
                                  "adjoint; %d stacked replicas/GPU" % (N, T - 1, R),
                      "replicas_per_gpu": R, "parallelism": "replica-dp%d" % world, "loss": float(loss.detach())}}
    if rank == 0:
        # roofline of the dominant kernel of this workload (profiles/r01f_schnet4096x8_kernel_stats.txt): the
        # edge-wise f32 GEMM [E,128] x [128,128] of the filter network and of its tangent / reverse sweeps,
        # issued through the library (hipBLASLt on the bucket-padded edge count, as mdgrad_amd/nn/analytic.py
        # does); f32 MFMA peak 157.3 TFLOP/s
        from mdgrad_amd.nn import analytic
        topo = analytic._stable(integ.model.models["gnn"].inputs["_topo"])
        E, F = topo.n_edges, 128
        A_ = torch.randn(E, F, device=dev)
        W_ = torch.randn(F, F, device=dev)
        with torch.no_grad(), analytic._blas_for(topo):
            A_.mm(W_)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                A_.mm(W_)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        flop = 2.0 * E * F * F
        out["roofline"] = {"bound": "mfma", "kernel": "library f32 GEMM [E,128]x[128,128] (hipBLASLt)",
                           "achieved": flop / (ms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                           "frac": flop / (ms * 1e-3) / 1e12 / 157.3, "traffic": None, "kernel_ms": ms,
                           "note": "E=%d padded edges; each row reads 512 B of A and writes 512 B of C for 32 768 flop: "
                                   "%.0f GB/s of operand traffic alongside the MFMA rate" % (E, 8.0 * E * F / ms / 1e6)}
        print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="lj108", choices=["lj108", "schnet4096"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--replicas", type=int, default=None, help="replicas per GPU (default 16384 for lj108, 8 for schnet4096)")
    ap.add_argument("--frames", type=int, default=None, help="saved frames T (T-1 MD steps); default 50 / 11")
    ap.add_argument("--dt", type=float, default=0.005)
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from mdgrad_amd import dist as mdist, ops, _lib
    from mdgrad_amd import potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    from mdgrad_amd.observable import rdf
    from mdgrad_amd.system import System

    rank, world, dev = mdist.init()
    if dev.type != "cuda":
        raise SystemExit("bench.py needs a HIP device (the hot path has no CPU implementation)")
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if args.workload == "schnet4096":
        args.replicas = 8 if args.replicas is None else args.replicas
        args.frames = 11 if args.frames is None else args.frames
        schnet_workload(args, rank, world, dev, mdist)
        import torch.distributed as tdist
        if tdist.is_available() and tdist.is_initialized():
            mdist.barrier()
            tdist.destroy_process_group()
        return
    args.replicas = 16384 if args.replicas is None else args.replicas
    args.frames = 50 if args.frames is None else args.frames
    R, T = args.replicas, args.frames
    atoms, pos, vel = make_inputs(R, 1000 + rank, dev)
    system = System(atoms, device=dev)
    mdl = P.LennardJones(1.0, 1.0)
    integ = NoseHooverChain(Stack({"pair": PairPotentials(system, mdl, cutoff=2.5)}), system, T=1.0,
                            num_chains=5, Q=50.0).to(dev)
    obs = rdf(system, nbins=100, r_range=(0.75, 2.5))
    target = torch.ones(100, device=dev)
    t = torch.Tensor([args.dt * i for i in range(T)]).to(dev)
    pv0 = torch.zeros(R, 5, device=dev)
    spec = integ.fused_spec("NH_verlet")
    spec.block = args.block
    params = list(integ.parameters())
    opt = torch.optim.Adam(params, lr=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        theta = spec.flat_params()
        v_t, q_t, pv_t = ops.FusedTrajFn.apply(vel, pos, pv0, t, theta, spec)
        _, _, g = obs(q_t)
        loss = (g - target).pow(2).mean()
        loss.backward()
        mdist.all_reduce_grads(params)           # the one collective per outer step
        opt.step()
        step.last_q = q_t
        return loss

    for _ in range(args.warmup):
        step()
    mdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    mdist.barrier()
    el = mdist.max_over_ranks(time.perf_counter() - t0, dev)

    if not (bool(torch.isfinite(step.last_q).all()) and all(bool(torch.isfinite(p).all()) for p in params)):
        raise SystemExit("bench: non-finite trajectory or parameters -- the measurement would be invalid")
    md_steps = R * (T - 1) * world * args.steps
    out = {"metric": "MD steps/sec (fwd+adjoint), 108-atom LJ NHC", "value": md_steps / el,
           "unit": "MD steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "FCC 3x3x3 LJ(1,1) 108 atoms, cutoff 2.5, NoseHooverChain(Q=50, 5 chains) "
                                  "velocity-Verlet, %d steps fwd + RDF(100 bins) loss + adjoint; "
                                  "%d replicas/GPU per pass" % (T - 1, R),
                      "replicas_per_gpu": R, "md_steps_per_pass": R * (T - 1), "parallelism": "replica-dp%d" % world,
                      "loss": float(loss.detach())}}

    if rank == 0:
        # ---- roofline of the dominant kernel (adjoint sweep), timed with HIP events on the
        # launch stream over repeated launches of exactly that kernel
        import ctypes as C
        lib = _lib.load()
        theta = spec.flat_params().detach().contiguous()
        v_t, q_t, pv_t = [x.detach() for x in ops.FusedTrajFn.apply(vel, pos, pv0, t, theta, spec)]
        gq = torch.randn_like(q_t) * 1e-3
        adj = [torch.empty(R, 108, 3, device=dev), torch.empty(R, 108, 3, device=dev),
               torch.empty(R, 5, device=dev), torch.zeros(R, spec.n_theta_total, device=dev)]
        prm = spec.params(R, T)

        def adj_launch():
            _lib.check(lib.mdg_traj_adj_small(C.byref(prm), C.byref(spec.cell_struct), C.byref(spec.terms),
                                              _lib.ptr(theta), _lib.ptr(spec.mass), _lib.ptr(t), _lib.ptr(v_t),
                                              _lib.ptr(q_t), _lib.ptr(pv_t), None, _lib.ptr(gq), None,
                                              _lib.ptr(adj[0]), _lib.ptr(adj[1]), _lib.ptr(adj[2]),
                                              _lib.ptr(adj[3]), _lib.stream_ptr(dev)), "adj")
        adj_launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            adj_launch()
        e1.record()
        torch.cuda.synchronize()
        adj_ms = e0.elapsed_time(e1) / reps
        # phase breakdown of one pass (HIP events on the launch stream): SURVEY 8d defines the metric on
        # t_fwd + t_adjoint with the RDF reported separately; `value` above is the stricter whole-pass rate
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        opt.zero_grad(set_to_none=True)
        ev[0].record()
        v2, q2, p2 = ops.FusedTrajFn.apply(vel, pos, pv0, t, spec.flat_params(), spec)
        ev[1].record()
        l2 = (obs(q2)[2] - target).pow(2).mean()
        ev[2].record()
        l2.backward()
        ev[3].record()
        torch.cuda.synchronize()
        fwd_ms, rdf_ms, bwd_ms = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3])
        out["config"]["phase_ms"] = {"traj_fwd": fwd_ms, "rdf_fwd": rdf_ms, "rdf_bwd_plus_traj_adj": bwd_ms,
                                     "traj_adj_kernel": adj_ms}
        out["config"]["md_steps_per_s_traj_only_per_gpu"] = R * (T - 1) / ((fwd_ms + adj_ms) * 1e-3)
        ell = ops.build_ell(pos[0], spec.cell_struct, 2.5)
        Pn = int(ell.half_list()[0].shape[0])
        N = 108
        bytes_adj = (48 * Pn + 208 * N) * (T - 1) * R           # 2 B_H + B_A + 2 B_N per step
        # HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic.json: FETCH_SIZE and
        # WRITE_SIZE collected in separate rocprofv3 --pmc runs, FETCH_SIZE doubled per the gfx950 note),
        # recorded per replica and scaled to this run's replica count
        traffic = None
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["traj_adj_kernel"]
            if pj["frames"] == T:
                traffic = pj["hbm_bytes_per_replica"] * R
        except Exception:
            traffic = None
        out["roofline"] = {"bound": "hbm", "kernel": "traj_adj_kernel", "achieved": bytes_adj / (adj_ms * 1e-3) / 1e9,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": bytes_adj / (adj_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                           "kernel_ms": adj_ms, "algorithmic_bytes_per_launch": bytes_adj,
                           "note": "algorithmic bytes of the unfused op chain (SURVEY 8d: 48P+208N per adjoint "
                                   "step, P=%d); the fused kernel keeps state in LDS so real HBM traffic is "
                                   "far lower and the kernel is VALU-bound" % Pn}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(T, args.dt)
        print(json.dumps(out))
    import torch.distributed as tdist
    if tdist.is_available() and tdist.is_initialized():
        mdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
