#!/usr/bin/env python
"""Benchmark of the differentiable-MD hot path on N MI355X GPUs: MD steps/sec (forward + adjoint).

Headline (BASELINE.json configs[1]): the 108-atom LJ system -- R independent replicas x (T-1) NH-Verlet steps
forward (one fused launch), the soft-histogram RDF loss and its gradient, the full adjoint sweep (one fused
launch), the single all-reduce of the parameter gradient (N > 1) and the optimizer step.  One "step" of this
script = one such pass; value = R*(T-1)*N_gpus*K / time.  Inputs are resident in HBM before the timed region.

The same JSON line nests, under "secondary", the two other north-star workloads measured the same way (each
with its own value / ms_per_step / roofline / cpu_baseline):
  schnet4096   4 096-bead CG water, SchNet A64/F128/G30/2 conv + ExcludedVolume prior, 8 stacked replicas / GPU, 52-step passes
  lj4096       4 096-atom LJ liquid (BASELINE config #4), fused large-N kernels, 64 replicas / GPU

    python bench.py --gpus 1 --steps 10 --warmup 2
    python bench.py --gpus 8 --steps 10 --warmup 2         # spawns its own 8 ranks (torch.distributed.run, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29511 bench.py --gpus 8 --steps 10 --warmup 2
    python bench.py --workload schnet4096      # one workload alone as the printed line
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
VEC_F32_PEAK_TF = 157.3      # MI355X_MICROARCH.md: fp32 vector (packed v_pk_fma_f32) = fp32 MFMA peak, TFLOP/s
MFMA_F32_PEAK_TF = 157.3
MFMA_BF16_PEAK_TF = 2500.0     # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
# arithmetic of one directed pair in the adjoint's fused sweep (force + Hessian.w + d/dtheta, csrc/traj_small.hip
# force_lj126_packed LEVEL 2): minimum image 12, d^2 5, 1/d^2 and the even-power polynomial 14, force 6, w-difference
# and projections 11, Hessian terms 14, theta sums 8  ~= 70 flop
FLOP_PER_PAIR_ADJ = 70.0
# the ring sweep (csrc/traj_ring.hpp) evaluates an undirected pair ONCE and updates both ends: the 70 flop above
# plus the visitor's accumulators (3 fma + 3 sub = 9 flop)
FLOP_PER_PAIR_RING = 79.0


# the f32 SchNet legs: every sum, activation and product is f32; the FORWARD filter sweeps (n_gaussians <= 32) form each product
# of their two Dense layers from three exact bf16 pieces per operand on the bf16 matrix pipe -- six piece products, error
# 1.8e-7 of sum |terms| against 1.9e-7 for v_mfma_f32_16x16x4_f32 (profiles/r06_split_mfma_accuracy.txt; MDG_F32_X6=0: the f32
# matrix instruction)
F32_SCHNET = "f32 (forward filter products as six exact-bf16-piece products: f32-accurate)"

def _measured_vector_peak():
    """TFLOP/s of back-to-back v_pk_fma_f32 on this chip as tools/micro/valu_rate.hip measured it (profiles/r05_valu_rate.txt):
    a packed fma issues in ~5.3 cycles per SIMD against ~2.9 for a plain v_fma_f32, so the 157.3 TF of the data sheet (one
    packed fma per SIMD every 4 cycles) is not reachable by any instruction stream; 118 TF is."""
    try:
        for ln in open(os.path.join(ROOT, "profiles", "r05_valu_rate.txt")):
            if ln.startswith("v_pk_fma_f32"):
                return float(ln.split()[-2])
    except (OSError, ValueError, IndexError):
        pass
    return None


def _events(n):
    return [torch.cuda.Event(enable_timing=True) for _ in range(n)]


def _finite(*ts):
    return all(bool(torch.isfinite(t).all()) for t in ts)


def _profile_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


# which source files a kernel's code comes from: counters are refused when one of them changed since they were collected
_KERNEL_SOURCES = (("traj_", ("traj_ring.hpp", "traj_small.hip", "common.hpp")), ("large_", ("traj_large.hip", "common.hpp")),
                   ("rdf_cell", ("rdf_cell.hip", "common.hpp")), ("rdf_", ("rdf.hip", "common.hpp")),
                   ("cfconv_", ("cfconv_fused.hip",)), ("dense_", ("dense.hip",)), ("grad_", ("gradjobs.hip",)),
                   ("nbr_", ("nbr.hip", "common.hpp")), ("row_chain", ("rowchain.hip",)), ("nhv_", ("nhc.hip", "common.hpp")))


def _source_sha(fname):
    import hashlib
    try:
        return hashlib.sha256(open(os.path.join(ROOT, "mdgrad_amd", "csrc", fname), "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def _counters(workload, kernel_prefix):
    """The rocprofv3 counter record (profiles/pmc_<workload>.json, tools/pmc_collect.py) of the kernel whose name starts
    with `kernel_prefix` -- the instance with the largest share of the pass -- or (None, reason): no record, or a record
    collected from different kernel sources than the ones in the tree (stamped per file)."""
    pj = _profile_json("pmc_%s.json" % workload)
    if not pj:
        return None, "no profiles/pmc_%s.json" % workload
    hit = [(k, r) for k, r in pj["kernels"].items() if k.startswith(kernel_prefix)]
    if not hit:
        return None, "kernel %s not in profiles/pmc_%s.json" % (kernel_prefix, workload)
    name, rec = max(hit, key=lambda kr: kr[1]["share"])
    for pre, files in _KERNEL_SOURCES:
        if name.startswith(pre):
            stale = [f for f in files if pj.get("sources", {}).get(f) != _source_sha(f)]
            if stale:
                return None, "stale counters: %s changed since profiles/pmc_%s.json was collected" % (", ".join(stale), workload)
            break
    rec = dict(rec, name=name)
    return rec, None


def _pass_traffic(workload, passes):
    """HBM bytes per pass summed over every profiled kernel (calls x bytes per launch / passes of the profiled run)."""
    pj = _profile_json("pmc_%s.json" % workload)
    if not pj:
        return None
    tot = sum(r["calls"] * r["hbm_bytes_per_launch"] for r in pj["kernels"].values() if "hbm_bytes_per_launch" in r)
    return tot / passes


def _dist_record(mdist, dev, params, ms_rank):
    """Evidence that the N ranks ran and met in the collective: backend, ranks counted by an
    all-reduce of ones, the time of one all-reduce of the flat-gradient-sized buffer, every rank's ms per pass."""
    rec = mdist.collective_evidence(dev, sum(p.numel() for p in params))
    rec["per_rank_ms"] = mdist.gather_over_ranks(ms_rank, dev)
    # every rank applied the same reduced gradient in the same optimizer: its parameters are bit-identical to rank 0's
    sums = mdist.gather_over_ranks(float(sum(p.detach().double().sum() for p in params)), dev)
    rec["param_checksum"] = sums[0]
    rec["params_identical_on_all_ranks"] = all(x == sums[0] for x in sums)
    return rec


# CPU legs (oracle baselines, parity of the timed geometry against the oracle) of a run over all workloads are collected here
# and executed AFTER the last timed GPU leg: measured in round 4 -- with the oracle's 16-thread CPU work between the legs the
# stacked SchNet pass took 27.6 ms instead of 26.3 and the lj4096 pass 22.9 instead of 20.9 (the same kernels, a slower host
# / device right after the CPU burst); nothing about the legs themselves depends on the order.
_DEFERRED = None


def _later(fn):
    if _DEFERRED is not None:
        _DEFERRED.append(fn)
    else:
        fn()


class _PassTrace:
    """MDG_BENCH_TRACE=1: host time of every timed pass (the launches stay asynchronous: a pass that ends in a host sync, as
    the SchNet passes do, shows its full time) and every collection of the cyclic collector, on stderr.  Diagnostics only."""

    def __init__(self, name):
        self.on = os.environ.get("MDG_BENCH_TRACE") == "1"
        self.name, self.t, self.gcs, self._t0 = name, [], [], 0.0
        if self.on:
            import gc
            gc.callbacks.append(self._gc)

    def _gc(self, phase, info):
        if phase == "start":
            self._g0 = time.perf_counter()
        else:
            self.gcs.append((len(self.t), info["generation"], round((time.perf_counter() - self._g0) * 1e3, 2)))

    def tick(self):
        if self.on:
            now = time.perf_counter()
            self.t.append(round((now - self._t0) * 1e3, 2))
            self._t0 = now

    def start(self):
        self._t0 = time.perf_counter()

    def done(self):
        if self.on:
            import gc
            gc.callbacks.remove(self._gc)
            print("[trace %s] ms per pass: %s\n[trace %s] collections (pass, generation, ms): %s" % (
                self.name, self.t, self.name, [g for g in self.gcs if g[2] >= 1.0]), file=sys.stderr, flush=True)


def _host_cpus():
    """(os.cpu_count(), CPUs this process may actually use: the cgroup quota / affinity mask when one is set)."""
    total = os.cpu_count() or 1
    usable = total
    try:
        usable = min(usable, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            usable = min(usable, max(1, int(round(float(q) / float(per)))))
    except (OSError, ValueError):
        pass
    return total, usable


def _cpu_timed(one, steps_per_call, what, batch_s=2.5, tries=(1, 8, 16)):
    """SURVEY 8d's CPU-baseline protocol on the oracle: thread count chosen by one calibration call each (the tensors of a
    108-atom system are too small for one thread per core; the count that runs fastest is used and stated), then 1 warm-up
    and the MEDIAN of 5 timed batches of ~batch_s each.  -> the cpu_baseline record (kind "port")."""
    total, usable = _host_cpus()
    cand = sorted({max(1, min(k, usable)) for k in tries})
    best = None
    for k in cand:
        torch.set_num_threads(k)
        if best is None:
            one()                                    # first call of all: lazy initialisation, not timed
        t0 = time.perf_counter()
        one()
        el = time.perf_counter() - t0
        if best is None or el < best[1]:
            best = (k, el)
    threads, t_one = best
    torch.set_num_threads(threads)
    per_batch = max(1, int(round(batch_s / t_one)))
    one()                                            # warm-up at the chosen thread count
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(per_batch):
            one()
        times.append(time.perf_counter() - t0)
    med = sorted(times)[2]
    return {"value": per_batch * steps_per_call / med, "unit": "MD steps/s", "cores": threads, "threads": threads,
            "host_cores": total, "usable_cores": usable, "threads_tried": cand, "kind": "port",
            "protocol": "1 warm-up + median of 5 batches", "batch_s": [round(x, 3) for x in times],
            "sample": "5 batches x %d trajectories x %d steps of %s, oracle/ (CPU restatement of the reference, pinned to its "
                      "golden vectors) on %d torch threads (fastest of %s; host: %d cores, %d usable by this process), "
                      "median batch %.2f s" % (per_batch, steps_per_call, what, threads, cand, total, usable, med)}


# ====================================================================================== 108-atom LJ (headline)
def make_inputs(R, seed, dev):
    from mdgrad_amd.system import FaceCenteredCubic
    rng = np.random.default_rng(seed)
    atoms = FaceCenteredCubic(symbol="H", size=(3, 3, 3), latticeconstant=1.6)
    lat = atoms.get_positions()
    pos = np.mod(lat[None] + rng.uniform(-0.05, 0.05, (R,) + lat.shape), 4.8).astype(np.float32)
    vel = rng.normal(0.0, np.sqrt(1.0 / 1.008), pos.shape).astype(np.float32)
    return atoms, torch.from_numpy(pos).to(dev), torch.from_numpy(vel).to(dev)


def _oracle_lj108(pos, vel, frames, dt, O, form="lj"):
    cell = torch.tensor([4.8] * 3)
    t = torch.Tensor([dt * i for i in range(frames)])
    term = (O.PairTerm("lj", torch.tensor([1.0, 1.0]), 2.5, cell, p=12, q=6, c=1) if form == "lj" else
            O.PairTerm("lj", torch.tensor([1.0, 1.0]), 2.5, cell, p=12, q=0, c=0))       # ExcludedVolume(sigma 1, epsilon 1, power 12)
    eom = O.NHCOracle(O.ModelOracle([term]), torch.full((108,), 1.008), 1.0, 50.0, 5)
    traj = O.odeint_oracle(eom, (vel, pos, torch.zeros(5)), t)
    leaves = [x.clone().requires_grad_(True) for x in traj]
    _, _, g = O.rdf_oracle(leaves[1], cell, 100, (0.75, 2.5))
    (g - 1).pow(2).mean().backward()
    lam, gth = O.adjoint_oracle(eom, traj, [x.grad for x in leaves], t)
    return traj, g.detach(), gth


def cpu_baseline_lj108(frames, dt, check=None, budget_s=12.0, form="lj", timed=True):
    """The CPU oracle (port of the reference algorithm, oracle/) on this host: the same 108-atom workload, one
    replica at a time, forward + rdf loss + adjoint; bounded to ~budget_s.  `check` = [(r, pos_r, vel_r, q_t_r, g_r,
    gth_r, R)] for sampled replicas r of the timed launch: their HIP results are compared with the oracle's on the same
    inputs."""
    import oracle as O
    _, pos, vel = make_inputs(1, 123, "cpu")
    out = _cpu_timed(lambda: _oracle_lj108(pos[0], vel[0], frames, dt, O, form), frames - 1,
                     "the timed workload itself (108-atom %s, fwd + rdf loss + adjoint, one replica at a time)" % form,
                     batch_s=2.5 if timed else 0.6)
    if check:
        dq = dg = dth = 0.0
        for (r, p0, v0, q_hip, g_hip, gth_hip, n_rep) in check:
            traj, g_o, gth_o = _oracle_lj108(p0, v0, frames, dt, O, form)
            dq = max(dq, float((q_hip - traj[1]).abs().max()))
            dg = max(dg, float((g_hip - g_o).abs().max()))
            dth = max(dth, float(((gth_hip - gth_o).abs() / gth_o.abs().max()).max()))
        out["parity_sampled"] = {
            "replicas": [c[0] for c in check], "max_abs_dq": dq, "max_abs_dg": dg, "rel_dtheta": dth,
            "note": "first / middle / last replica of a %s-replica launch (the timed geometry) vs the oracle on the same "
                    "inputs, worst of the three: positions over %d frames, g(r) of that replica, d(loss)/d(sigma, epsilon)"
                    % (check[0][6], frames)}
    return out


def _lj108_compulsory_bytes(N, R, T, n_theta, chains=5):
    """HBM bytes the fused adjoint launch cannot avoid: T saved frames of (v, q) [N, 3] f32 and the thermostat momenta per replica in,
    the three costates and the parameter-gradient row per replica out.  (The forward launch writes the same frames: a pass
    moves about twice this.)"""
    frames_in = R * T * (2 * N * 3 + chains) * 4
    out = R * (2 * N * 3 + chains + n_theta) * 4
    return frames_in + out


def _ordered_config(cfg):
    """The headline's `config` with the keys the driver's parser must not lose FIRST (it keeps the first ~20 scalar keys and cuts
    strings at 100 characters): workload (<= 100 chars), then rate / time per pass / dominant-kernel
    roofline fraction of every other BASELINE config, then the parity and dtype scalars, then everything else."""
    first = ["workload", "replicas_per_gpu"]
    for name in ("lj4096", "water192", "schnet4096", "single_system"):
        first += [name + "_md_steps_per_s", name + "_ms_per_pass", name + "_kernel_frac"]
    first += ["exvol108_md_steps_per_s", "water192x64_md_steps_per_s", "rdf_fused"]
    out = {k: cfg[k] for k in first if k in cfg}
    if isinstance(out.get("workload"), str) and len(out["workload"]) > 100:
        out["workload_full"] = out["workload"]
        out["workload"] = out["workload"][:97] + "..."
    for k, v in cfg.items():
        if k not in out:
            out[k] = v
    return out


def run_lj108(args, rank, world, dev, mdist, with_cpu=True, form="lj", dt=None, steps=None, warmup=None):
    """form "lj": LennardJones(1, 1), the headline.  form "exvol": the reference README's own demo model,
    ExcludedVolume(sigma 1, epsilon 1, power 12) at its dt = 0.01 (README.md:70-85): the same ring kernels -- the even-power
    polynomial with the attractive coefficient 0."""
    import copy
    args = copy.copy(args)
    if dt is not None:
        args.dt = dt
    if steps is not None:
        args.steps = steps
    if warmup is not None:
        args.warmup = warmup
    import ctypes as C
    from mdgrad_amd import ops, _lib
    from mdgrad_amd import potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    from mdgrad_amd.observable import rdf
    from mdgrad_amd.system import System
    R = 16384 if args.replicas is None else args.replicas
    T = 50 if args.frames is None else args.frames
    atoms, pos, vel = make_inputs(R, 1000 + rank, dev)
    system = System(atoms, device=dev)
    mdl = P.LennardJones(1.0, 1.0) if form == "lj" else P.ExcludedVolume(1.0, 1.0, 12)
    integ = NoseHooverChain(Stack({"pair": PairPotentials(system, mdl, cutoff=2.5)}), system, T=1.0,
                            num_chains=5, Q=50.0).to(dev)
    obs = rdf(system, nbins=100, r_range=(0.75, 2.5))
    target = torch.ones(100, device=dev)
    t = torch.Tensor([args.dt * i for i in range(T)]).to(dev)
    pv0 = torch.zeros(R, 5, device=dev)
    integ.fuse_observables = True        # opt-in: the RDF of the selected frames is evaluated inside the trajectory launches
    spec = integ.fused_spec("NH_verlet")
    spec.block = args.block
    params = list(integ.parameters())
    opt = torch.optim.Adam(params, lr=1e-4)

    # ---- parity of the timed geometry (before any optimizer step): replica 0 alone feeds the loss
    check = []
    if rank == 0 and with_cpu and world == 1:
        for r in sorted({0, R // 2, R - 1}):           # one replica at a time feeds the loss of the whole launch
            v_t, q_t, pv_t = ops.fused_traj(vel, pos, pv0, t, spec.flat_params(), spec)
            _, _, g0 = obs(q_t[r:r + 1])
            (g0 - 1).pow(2).mean().backward()
            check.append((r, pos[r].cpu(), vel[r].cpu(), q_t[r].detach().cpu(), g0.detach().cpu(),
                          torch.stack([mdl.sigma.grad.reshape(()), mdl.epsilon.grad.reshape(())]).cpu(), R))
            opt.zero_grad(set_to_none=True)

    def step():
        opt.zero_grad(set_to_none=True)
        theta = spec.flat_params()
        v_t, q_t, pv_t = ops.fused_traj(vel, pos, pv0, t, theta, spec)
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
    el_rank = time.perf_counter() - t0
    el = mdist.max_over_ranks(el_rank, dev)
    if not (_finite(step.last_q) and all(_finite(p) for p in params)):
        raise SystemExit("bench: non-finite trajectory or parameters -- the measurement would be invalid")
    md_steps = R * (T - 1) * world * args.steps
    label = "LJ(1,1)" if form == "lj" else "ExcludedVolume(sigma 1, eps 1, power 12)"
    out = {"metric": "MD steps/sec (fwd+adjoint), 108-atom %s NHC" % ("LJ" if form == "lj" else "ExcludedVolume"), "value": md_steps / el,
           "unit": "MD steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "FCC 3^3 %s 108 atoms rc 2.5 NHC(Q50,5) dt %g, %d steps+RDF loss+adjoint, %d rep/GPU" % (
                          "LJ(1,1)" if form == "lj" else "ExVol(1,1,12)", args.dt, T - 1, R),
                      "workload_detail": "FCC 3x3x3 %s 108 atoms, cutoff 2.5, NoseHooverChain(Q=50, 5 chains) "
                                         "velocity-Verlet dt %g, %d steps fwd + RDF(100 bins) loss + adjoint; "
                                         "%d replicas/GPU per pass" % (label, args.dt, T - 1, R),
                      "replicas_per_gpu": R, "md_steps_per_pass": R * (T - 1), "parallelism": "replica-dp%d" % world,
                      "loss": float(loss.detach())}}
    out["config"]["dist"] = _dist_record(mdist, dev, params, el_rank / args.steps * 1e3)
    if rank != 0:
        return out
    # ---- roofline of the dominant kernel (adjoint sweep): HIP events on the launch stream over repeated launches
    lib = _lib.load()
    theta = spec.flat_params().detach().contiguous()
    v_t, q_t, pv_t = [x.detach() for x in ops.FusedTrajFn.apply(vel, pos, pv0, t, theta, spec)]
    gq = torch.randn_like(q_t) * 1e-3
    adj = [torch.empty(R, 108, 3, device=dev), torch.empty(R, 108, 3, device=dev),
           torch.empty(R, 5, device=dev), torch.zeros(R, spec.n_theta_total, device=dev)]
    prm = spec.params(R, T)

    fuse = getattr(spec, "rdf_hint", None)           # (set by the observable during the warm-up passes)
    prm_ = spec.params(R, T)
    fused = bool(fuse is not None and lib.mdg_traj_rdf_supported(C.byref(prm_), C.byref(spec.cell_struct),
                                                                 C.byref(spec.terms), C.byref(fuse.struct())))
    g_raw = torch.randn(100, device=dev) * 1e-6

    def adj_launch():
        # the launch of the timed pass: with the RDF observable fused in, its frame gradients are produced inside
        if fused:
            _lib.check(lib.mdg_traj_adj_small_rdf(C.byref(prm), C.byref(spec.cell_struct), C.byref(spec.terms),
                                                  _lib.ptr(theta), _lib.ptr(spec.mass), _lib.ptr(t), _lib.ptr(v_t),
                                                  _lib.ptr(q_t), _lib.ptr(pv_t), None, None, None,
                                                  _lib.ptr(adj[0]), _lib.ptr(adj[1]), _lib.ptr(adj[2]),
                                                  _lib.ptr(adj[3]), C.byref(fuse.struct()), _lib.ptr(g_raw),
                                                  _lib.stream_ptr(dev)), "adj")
            return
        _lib.check(lib.mdg_traj_adj_small(C.byref(prm), C.byref(spec.cell_struct), C.byref(spec.terms),
                                          _lib.ptr(theta), _lib.ptr(spec.mass), _lib.ptr(t), _lib.ptr(v_t),
                                          _lib.ptr(q_t), _lib.ptr(pv_t), None, _lib.ptr(gq), None,
                                          _lib.ptr(adj[0]), _lib.ptr(adj[1]), _lib.ptr(adj[2]),
                                          _lib.ptr(adj[3]), _lib.stream_ptr(dev)), "adj")
    adj_launch()
    e0, e1 = _events(2)
    reps = 5
    e0.record()
    for _ in range(reps):
        adj_launch()
    e1.record()
    torch.cuda.synchronize()
    adj_ms = e0.elapsed_time(e1) / reps
    # phase breakdown of one pass: SURVEY 8d defines the metric on t_fwd + t_adjoint with the RDF reported
    # separately; `value` above is the stricter whole-pass rate
    ev = _events(4)
    opt.zero_grad(set_to_none=True)
    ev[0].record()
    v2, q2, p2 = ops.fused_traj(vel, pos, pv0, t, spec.flat_params(), spec)
    ev[1].record()
    l2 = (obs(q2)[2] - target).pow(2).mean()
    ev[2].record()
    l2.backward()
    ev[3].record()
    torch.cuda.synchronize()
    fwd_ms, rdf_ms, bwd_ms = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3])
    out["config"]["phase_ms"] = {"traj_fwd": fwd_ms, "rdf_fwd": rdf_ms, "rdf_bwd_plus_traj_adj": bwd_ms,
                                 "traj_adj_kernel": adj_ms}
    out["config"]["rdf_fused_into_trajectory_kernels"] = fused   # then traj_fwd holds the histogram, traj_adj its gradient
    # (said in the record, not only in a Python warning -- VERDICT r5 weak #1b: the fused observable is an approximation the
    #  reference does not make, inside the stated 1e-4 tolerance on g(r))
    out["config"]["rdf_fused"] = ("fine-grid histogram + cubic-Hermite gradient table, <=2e-5/bin vs the exact kernel" if fused
                                  else "off: exact rdf kernels")
    out["config"]["md_steps_per_s_traj_only_per_gpu"] = R * (T - 1) / ((fwd_ms + adj_ms) * 1e-3)
    ell = ops.build_ell(pos[0], spec.cell_struct, 2.5)
    Pn = int(ell.half_list()[0].shape[0])
    N = 108
    intervals = (T - 1) * R
    # The fused kernel keeps the state in registers for the whole sweep, so its binding roof is VALU issue, not HBM
    # (VERDICT r1 #6).  achieved = arithmetic of the UNDIRECTED pairs inside the cutoff, each evaluated once with
    # both ends updated (Newton's third law ring, csrc/traj_ring.hpp), two evaluations per interval; executed =
    # every pair slot the ring issues: per evaluation 1 + 2 (nl-1)/2 (+ 2 for even nl) packed operations of
    # 64 lanes x 2 pairs, idle lanes and pairs beyond the cutoff included.
    nl = (N + 1) // 2
    ring_ops = 1 + 2 * ((nl - 1) // 2) + (2 if nl % 2 == 0 else 0)
    useful = FLOP_PER_PAIR_RING * Pn * 2.0 * intervals
    executed = FLOP_PER_PAIR_RING * ring_ops * 128.0 * 2.0 * intervals
    bytes_adj = (48 * Pn + 208 * N) * intervals                    # 2 B_H + B_A + 2 B_N per step
    kname = "traj_adj_ring_kernel"
    # counters of the same launch (16 384 replicas, 50 frames, observable fused in), refused when the kernel source changed
    cnt, why = (_counters("lj108", "traj_adj_ring_kernel<true") if (R == 16384 and T == 50 and fused and form == "lj")
                else (None, "other geometry / pair form"))
    traffic = cnt.get("hbm_bytes_per_launch") if cnt else None
    sec = adj_ms * 1e-3
    out["roofline"] = {
        "bound": "valu", "kernel": kname, "achieved": useful / sec / 1e12, "peak": VEC_F32_PEAK_TF,
        "unit": "TFLOP/s", "frac": useful / sec / 1e12 / VEC_F32_PEAK_TF, "traffic": traffic, "kernel_ms": adj_ms,
        "executed_tflops": executed / sec / 1e12, "executed_frac": executed / sec / 1e12 / VEC_F32_PEAK_TF,
        "peak_measured": _measured_vector_peak(),
        "executed_frac_of_measured_peak": (executed / sec / 1e12 / _measured_vector_peak()) if _measured_vector_peak() else None,
        "valu_busy": cnt.get("valu_busy") if cnt else None, "wait_frac": cnt.get("wait_frac") if cnt else None,
        "counters": ("profiles/pmc_lj108.json: %s, %.1f us under rocprofv3" % (cnt["name"], cnt["avg_us"])) if cnt else why,
        "hbm_frac_measured": (traffic / sec / 1e9 / HBM_PEAK_GBS) if traffic else None,
        "hbm_frac_algorithmic": None,
        # what the FUSED design has to move per adjoint launch: the saved frames (v, q of 108 atoms + 5 thermostat momenta per
        # step-replica) come in once; the costates and the parameter gradient go out once per replica
        "hbm_compulsory_bytes_per_launch": _lj108_compulsory_bytes(N, R, T, spec.n_theta_total),
        "hbm_frac_compulsory": _lj108_compulsory_bytes(N, R, T, spec.n_theta_total) / sec / 1e9 / HBM_PEAK_GBS,
        "hbm_algorithmic_model": "void for this kernel: SURVEY 8d's bytes of the UNFUSED op chain (48P+208N per step) over the "
                                 "kernel time are %.2f x the 8 TB/s peak (> 1) -- the state lives in registers and those bytes "
                                 "never move; the roof that binds is VALU issue (frac above)" % (
                                     bytes_adj / sec / 1e9 / HBM_PEAK_GBS),
        "algorithmic_bytes_per_launch": bytes_adj,
        "note": "useful = %.0f flop x P = %d undirected pairs inside the cutoff (each evaluated once, both ends updated) x 2 "
                "evaluations x %d intervals; executed_* = the %d packed pair operations x 128 slots the ring issues per "
                "evaluation (N(N-1)/2 = %d pairs, 10 of 64 lanes own no atom).  The measured HBM traffic (frame + "
                "frame-gradient loads, profiles/pmc_lj108.json) is ~40x below SURVEY 8d's unfused-chain bytes" % (
                    FLOP_PER_PAIR_RING, Pn, intervals, ring_ops, N * (N - 1) // 2)}
    if with_cpu and world == 1:
        def cpu_part():
            out["cpu_baseline"] = cpu_baseline_lj108(T, args.dt, check, form=form, timed=form == "lj")
        _later(cpu_part)
    return out


# ====================================================================================== SchNet 4096 beads
def schnet_flops_forward(N, E, A, F, G, n_conv):
    """flops of one forward energy evaluation (E = undirected edges)."""
    per_conv = 2.0 * E * G * (G + F) + 2.0 * N * A * F + 4.0 * E * F + 2.0 * N * A * (F + A)
    return n_conv * per_conv + N * A * (A + 1)


def cpu_baseline_schnet(budget_s=10.0):
    """oracle/ SchNet path (autograd double backward like the reference) on a 64-bead CG-water box with the same
    network widths: forward + RDF loss + adjoint, bounded."""
    import oracle as O
    from mdgrad_amd import units
    from mdgrad_amd.nn import get_model
    rng = np.random.default_rng(7)
    a = units.get_unit_len(0.997, 18.01528, 8)
    pos, cell = O.diamond_lattice(2, a)
    pos = np.mod(pos + rng.normal(0, 0.05, pos.shape), cell).astype(np.float32)
    kT = 298.0 * units.kB
    vel = (rng.normal(0, 1, pos.shape) * np.sqrt(kT / 18.01528)).astype(np.float32)
    torch.manual_seed(0)
    net = get_model({"n_atom_basis": 64, "n_filters": 128, "n_gaussians": 30, "n_convolutions": 2, "cutoff": 6.0})
    with torch.no_grad():
        net.atomwisereadout.readout["energy"][2].weight.mul_(0.02)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    cellt = torch.tensor(cell, dtype=torch.float32)
    nsteps = 3
    t = torch.Tensor([units.fs * i for i in range(nsteps + 1)])

    def one():
        gnn = O.SchNetTerm(sd, np.full(len(pos), 8), 6.0, cellt)
        prior = O.PairTerm("lj", torch.tensor([2.6, 0.01]), 6.0, cellt, p=12, q=0, c=0)
        eom = O.NHCOracle(O.ModelOracle([gnn, prior]), torch.full((len(pos),), 18.01528), kT, 50.0, 5)
        traj = O.odeint_oracle(eom, (torch.from_numpy(vel), torch.from_numpy(pos), torch.zeros(5)), t)
        leaves = [x.clone().requires_grad_(True) for x in traj]
        _, _, g = O.rdf_oracle(leaves[1], cellt, 60, (2.0, 6.0))
        (g - 1).pow(2).mean().backward()
        O.adjoint_oracle(eom, traj, [x.grad for x in leaves], t)

    out = _cpu_timed(one, nsteps, "a 64-bead CG-water box with the timed SchNet widths (fwd + rdf loss + adjoint)")
    out["sample"] += ("; NOT the timed geometry: 8 x 4096 beads do not finish on a CPU in the bench's budget -- the oracle's cost "
                      "at 4096 beads is in parity_sampled.oracle_s_per_md_step; BASELINE.md: the reference ran 8.6 steps/s at 64 "
                      "beads, 3.9 at 512")
    return out


def build_schnet_workload(dev, R, bf16, seed, size=8, widths=(64, 128, 30, 2), rows16=False):
    """The SchNet workload of BASELINE config #5 as every leg of this script (and tests/test_gpu_secondary_pins.py) builds it:
    CG water on a Diamond size^3 lattice (8 size^3 beads, rho = 0.997 g/cm3), jittered and thermalised at 298 K with
    default_rng(seed), R replicas stacked into one system, SchNet(A, F, G, n_conv) with torch.manual_seed(0) weights (readout
    scaled by 0.02 so that the synthetic dynamics stay stable) + ExcludedVolume(2.6, 0.01, 12) prior, cutoff 6,
    NoseHooverChain(Q = 50, 5 chains).  bf16: bf16 MFMA operands in the filter network; rows16 (with bf16): the convolution
    kernels gather bf16 mirrors of the node matrices as well (SchNet.node_rows_bf16)."""
    from mdgrad_amd import potentials as P, units
    from mdgrad_amd.interface import GNNPotentials, PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    from mdgrad_amd.nn import get_model
    from mdgrad_amd.system import System, Diamond
    A_, F_, G_, NC = widths
    rng = np.random.default_rng(seed)
    a = units.get_unit_len(0.997, 18.01528, 8)
    atoms = Diamond("O", (size,) * 3, a)
    atoms.masses[:] = 18.01528
    base = System(atoms, device=dev)
    system = base.replicate(R) if R > 1 else base
    L = a * size
    system.set_positions(np.mod(system.get_positions() + rng.normal(0, 0.05, (len(system), 3)), L))
    kT = 298.0 * units.kB
    system.set_temperature(kT, rng=rng)
    torch.manual_seed(0)
    net = get_model({"n_atom_basis": A_, "n_filters": F_, "n_gaussians": G_, "n_convolutions": NC, "cutoff": 6.0})
    net.filter_bf16 = bool(bf16)
    net.node_rows_bf16 = bool(bf16 and rows16)
    with torch.no_grad():        # random-init SchNet forces are O(100 eV/A): scale the readout so the synthetic
        net.atomwisereadout.readout["energy"][2].weight.mul_(0.02)   # dynamics stay stable (checked by the callers)
    gnn = GNNPotentials(system, net, cutoff=6.0)
    integ = NoseHooverChain(Stack({"gnn": gnn, "prior": PairPotentials(system, P.ExcludedVolume(2.6, 0.01, 12), cutoff=6.0)}),
                            system, T=kT, num_chains=5, Q=50.0).to(dev)
    return dict(base=base, system=system, net=net, gnn=gnn, integ=integ, kT=kT, L=L, N=base.get_number_of_atoms(), R=R,
                widths=widths)


def schnet_oracle_replica(wl, sd, pos, vel, t, loss_fn):
    """One replica of a `build_schnet_workload` system on oracle/ (autograd double backward, like the reference):
    -> (trajectory, adjoints of the initial state, dL/dtheta)."""
    import oracle as O
    N, L = wl["N"], wl["L"]
    cellt = torch.tensor([L] * 3, dtype=torch.float32)
    gnn = O.SchNetTerm(sd, np.full(N, 8), 6.0, cellt)
    prior = O.PairTerm("lj", torch.tensor([2.6, 0.01]), 6.0, cellt, p=12, q=0, c=0)
    eom = O.NHCOracle(O.ModelOracle([gnn, prior]), torch.full((N,), 18.01528), wl["kT"], 50.0, 5)
    traj = O.odeint_oracle(eom, (torch.from_numpy(vel), torch.from_numpy(pos), torch.zeros(5)), t)
    leaves = [x.clone().requires_grad_(True) for x in traj]
    loss_fn(leaves).backward()
    lam, gth = O.adjoint_oracle(eom, traj, [x.grad if x.grad is not None else torch.zeros_like(x) for x in leaves], t)
    return traj, lam, gth


def parity_schnet_stacked(dev, bf16, R=8, size=2, T=11, stride=5, sample=None, seed=99, rows16=False):
    """The path the schnet4096 leg times (replica-stacked system, fused interaction block, analytic adjoint; HIP-graph replay
    up to 2^18 edges, the eager pass on stored lists beyond) against oracle/ (autograd double backward like the reference):
    R stacked replicas x 8 size^3 beads built exactly as the timed workload (`build_schnet_workload`), T - 1 steps +
    per-replica RDF loss + adjoint; the sampled replicas' trajectories and g(r), and the parameter gradient summed over them.
    size = 8, R = 8 IS the timed geometry (8 x 4 096 beads, 459 k edges)."""
    from mdgrad_amd import units
    from mdgrad_amd.observable import rdf
    from mdgrad_amd.sovlers import odeint_adjoint
    wl = build_schnet_workload(dev, R, bf16, seed, size=size, rows16=rows16)
    integ, system, base, N = wl["integ"], wl["system"], wl["base"], wl["N"]
    pos = system.get_positions().reshape(R, N, 3).astype(np.float32)
    vel = system.get_velocities().reshape(R, N, 3).astype(np.float32)
    sd = {k: v.detach().clone().cpu() for k, v in wl["net"].state_dict().items()}
    t = torch.Tensor([units.fs * i for i in range(T)])
    y0 = tuple(integ.get_inital_states(wrap=True))
    v_t, q_t, pv_t = odeint_adjoint(integ, y0, t.to(dev), method="NH_verlet")
    obs = rdf(base, nbins=60, r_range=(2.0, 6.0))
    sample = sorted({0, R // 2, R - 1}) if sample is None else list(sample)
    qr = q_t.reshape(T, R, N, 3)
    gs = [obs(qr[::stride, r])[2] for r in sample]
    sum((g - 1).pow(2).mean() for g in gs).backward()
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in integ.parameters()]).cpu()
    cellt = torch.tensor([wl["L"]] * 3, dtype=torch.float32)
    import oracle as O
    dq = dg = 0.0
    gsum = None
    total, usable = _host_cpus()
    threads = max(1, min(usable, 32))
    torch.set_num_threads(threads)
    t_or = time.perf_counter()
    for r, g in zip(sample, gs):
        traj, lam, gth = schnet_oracle_replica(
            wl, sd, pos[r], vel[r], t, lambda L_: (O.rdf_oracle(L_[1][::stride], cellt, 60, (2.0, 6.0))[2] - 1).pow(2).mean())
        go = O.rdf_oracle(traj[1][::stride], cellt, 60, (2.0, 6.0))[2]
        gsum = gth if gsum is None else gsum + gth
        dq = max(dq, float((qr[:, r].detach().cpu() - traj[1]).abs().max()))
        dg = max(dg, float((g.detach().cpu() - go.detach()).abs().max()))
    t_or = time.perf_counter() - t_or
    cos = float((flat.double() * gsum.double()).sum() / (flat.double().norm() * gsum.double().norm()))
    return {"replicas": sample, "beads_per_replica": N, "stacked_replicas": R, "steps": T - 1,
            "oracle_s_per_md_step": t_or / (len(sample) * (T - 1)), "oracle_threads": threads, "host_cores": total,
            "edges": int(wl["gnn"].inputs["_topo"].n_edges), "max_abs_dq": dq, "max_abs_dg": dg,
            "rel_dtheta": float((flat - gsum).abs().max() / gsum.abs().max()), "cos_dtheta": cos,
            "filter": ("bf16 MFMA operands + bf16 gathered node rows" if rows16 else "bf16 MFMA operands") if bf16 else "f32",
            "note": "%d stacked replicas x %d CG-water beads built as the timed launch, same SchNet widths, %d steps + "
                    "per-replica RDF loss + analytic adjoint: replicas %s vs oracle/ (positions in A over %d frames, g(r)), and "
                    "the %d-entry parameter gradient summed over them (largest deviation relative to the largest entry, "
                    "cosine); the same geometry is pinned in tests/test_gpu_secondary_pins.py" % (
                        R, N, T - 1, sample, T, flat.numel())}


def schnet_single_system(dev, bf16, T=21, passes=4, rows16=False):
    """The same model and loss on ONE 4 096-bead system (no replica stacking): the launch-bound end of the SchNet path --
    each MD step is ~75 graph nodes of 5-50 us.  -> MD steps/s over `passes` timed passes of T - 1 steps (forward +
    adjoint + RDF loss + optimizer step), after two warm-up passes."""
    from mdgrad_amd import units
    from mdgrad_amd.observable import rdf
    from mdgrad_amd.sovlers import odeint_adjoint
    wl = build_schnet_workload(dev, 1, bf16, 77, rows16=rows16)
    system, integ = wl["system"], wl["integ"]
    obs = rdf(system, nbins=60, r_range=(2.0, 6.0))
    target = torch.ones(60, device=dev)
    t = torch.Tensor([units.fs * i for i in range(T)]).to(dev)
    params = list(integ.parameters())
    opt = torch.optim.Adam(params, lr=1e-5)

    def step():
        opt.zero_grad(set_to_none=True)
        y0 = tuple(integ.get_inital_states(wrap=True))
        v_t, q_t, pv_t = odeint_adjoint(integ, y0, t, method="NH_verlet")
        (obs(q_t[::5])[2] - target).pow(2).mean().backward()
        opt.step()
        return q_t

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(passes):
        q_t = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if not _finite(q_t):
        return {"error": "non-finite trajectory"}
    return {"md_steps_per_s": passes * (T - 1) / el, "us_per_md_step": el / (passes * (T - 1)) * 1e6, "beads": len(system),
            "ms_per_pass": el / passes * 1e3,
            "filter": ("bf16 MFMA operands + bf16 gathered node rows" if rows16 else "bf16 MFMA operands") if bf16 else "f32",
            "note": "ONE 4096-bead system, %d passes x %d steps fwd + RDF loss + analytic adjoint + Adam step (HIP-graph replay "
                    "of the per-step launches; node-level layers as row chains, csrc/rowchain.hip); tools/gbench.py gnn4096 "
                    "times the same without the optimizer" % (passes, T - 1)}


def run_schnet4096(args, rank, world, dev, mdist, with_cpu=True, steps=None, warmup=None):
    from mdgrad_amd import ops, units
    from mdgrad_amd.observable import rdf
    from mdgrad_amd.sovlers import odeint_adjoint
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    R = 8 if args.replicas is None or args.workload != "schnet4096" else args.replicas
    # (opt_freq = 52 steps per pass; `frames_override` = 11 gives the 10-step passes of rounds 2-4)
    T = 53 if args.frames is None or args.workload != "schnet4096" else args.frames
    T = getattr(args, "frames_override", None) or T
    A_, F_, G_, NC = 64, 128, 30, 2
    rows16 = bool(args.bf16 and getattr(args, "bf16_rows", False))
    wl = build_schnet_workload(dev, R, args.bf16, 2000 + rank, widths=(A_, F_, G_, NC), rows16=rows16)
    base, system, net, gnn, integ = wl["base"], wl["system"], wl["net"], wl["gnn"], wl["integ"]
    obs = rdf(system, nbins=60, r_range=(2.0, 6.0))
    target = torch.ones(60, device=dev)
    t = torch.Tensor([units.fs * i for i in range(T)]).to(dev)
    params = list(integ.parameters())
    opt = torch.optim.Adam(params, lr=1e-5)

    # the initial state is uploaded ONCE (wrapped positions, velocities, thermostat momenta): every pass starts from the same
    # device-resident tensors, as the two Lennard-Jones legs do -- get_inital_states() wraps on the host and copies 0.8 MB to
    # the device, ~0.7 ms during which the GPU has nothing to do
    y0_dev = tuple(x.clone() for x in integ.get_inital_states(wrap=True))

    def step():
        opt.zero_grad(set_to_none=True)
        y0 = tuple(x.clone() for x in y0_dev)
        v_t, q_t, pv_t = odeint_adjoint(integ, y0, t, method="NH_verlet")
        loss = (obs(q_t[::5])[2] - target).pow(2).mean()
        loss.backward()
        mdist.all_reduce_grads(params)
        opt.step()
        return loss, q_t

    bf16_dev = None
    if args.bf16 and rank == 0:
        # deviation of the bf16-operand filter network from the all-f32 path ON THE TIMED WORKLOAD: one pass each from
        # the same initial state, no optimizer step in between
        def one_pass(flag):
            net.filter_bf16 = flag
            opt.zero_grad(set_to_none=True)
            y0 = tuple(integ.get_inital_states(wrap=True))
            v_t, q_t, pv_t = odeint_adjoint(integ, y0, t, method="NH_verlet")
            g = obs(q_t[::5])[2]
            (g - target).pow(2).mean().backward()
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
            return q_t.detach().clone(), g.detach().clone(), flat.double()
        qa, ga, fa = one_pass(False)
        qb, gb, fb = one_pass(True)
        opt.zero_grad(set_to_none=True)
        bf16_dev = {"max_abs_dq_A": float((qa - qb).abs().max()), "max_abs_dg": float((ga - gb).abs().max()),
                    "dtheta_rel_to_largest": float((fa - fb).abs().max() / fa.abs().max()),
                    "dtheta_cosine": float((fa * fb).sum() / (fa.norm() * fb.norm())),
                    "note": "bf16 filter operands%s vs all-f32 on the timed workload itself (%d steps, same initial state): "
                            "positions, g(r), the %d-entry parameter gradient" % (
                                " + bf16 mirrors of the gathered node rows" if rows16 else "", T - 1, fa.numel())}
    for _ in range(warmup):
        step()
    import gc
    gc.collect()
    gc.freeze()                 # (what exists now is permanent: the collector's sweeps inside the timed passes stay short)
    vl0 = (gnn._static or {}).get("verlet")
    builds0 = vl0.builds() if vl0 is not None else None
    mdist.barrier()
    torch.cuda.synchronize()
    tr = _PassTrace("schnet4096")
    t0 = time.perf_counter()
    tr.start()
    for _ in range(steps):
        loss, q_last = step()
        tr.tick()
    torch.cuda.synchronize()
    mdist.barrier()
    el_rank = time.perf_counter() - t0
    tr.done()
    el = mdist.max_over_ranks(el_rank, dev)
    if not (_finite(q_last) and all(_finite(p) for p in params)):
        raise SystemExit("bench: non-finite trajectory or parameters -- the measurement would be invalid")
    vl1 = (gnn._static or {}).get("verlet")
    nbr_info = None
    if vl1 is not None and vl1 is vl0:
        nbr_info = {"skin_A": vl1.skin, "searches_per_pass": (vl1.builds() - builds0) / float(steps),
                    "force_evaluations_per_pass": 3 * (T - 1) + 1,
                    "note": "stored list searched with cutoff + skin, kept while no bead has moved more than skin / 2 (device-side "
                            "decision); every evaluation re-applies the exact cutoff per pair (tests/test_gpu_verlet.py)"}
    N = base.get_number_of_atoms()
    md_steps = R * (T - 1) * world * steps
    out = {"metric": "MD steps/sec (fwd+adjoint), 4096-bead SchNet CG water NHC", "value": md_steps / el,
           "unit": "MD steps/s", "n_gpus": world, "steps": steps, "warmup": warmup,
           "ms_per_step": el / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": ("bf16 filter MFMA operands + bf16 mirrors of the gathered node rows, f32 products and accumulate" if rows16
                     else "bf16 filter MFMA operands, f32 accumulate") if args.bf16 else F32_SCHNET, "data": "synthetic",
           "config": {"workload": "CG water Diamond 8^3 (%d beads), SchNet A64 F128 G30 2 conv + ExcludedVolume prior, "
                                  "cutoff 6, NoseHooverChain(Q=50, 5 chains), %d steps fwd + RDF(60 bins) loss + analytic "
                                  "adjoint; %d stacked replicas/GPU" % (N, T - 1, R),
                      "replicas_per_gpu": R, "parallelism": "replica-dp%d" % world, "loss": float(loss.detach())}}
    out["config"]["dist"] = _dist_record(mdist, dev, params, el_rank / steps * 1e3)
    if bf16_dev is not None:
        out["config"]["bf16_vs_f32"] = bf16_dev
    if nbr_info is not None:
        out["config"]["neighbour_list"] = nbr_info
    if rank != 0:
        return out
    # ---- roofline: (a) the kernel with the largest share of the pass (profiles/*schnet4096_kernel_stats.txt): the reverse
    # sweep of the fused interaction block with parameter gradients, cfconv_bwd_kernel<32,8,true,true>, timed with HIP
    # events on the launch stream on the step's own topology; the forward + tangent sweep beside it; (b) the whole step's
    # MFMA-eligible flops (21 x forward) over the step time
    from mdgrad_amd.nn import analytic
    topo = gnn.inputs["_topo"]
    NN, E = topo.n_atoms, topo.n_edges
    conv = net.convolutions[0]
    Pm = analytic._layer_params(conv)
    fn = ops.FilterNet(Pm["mu"], Pm["c"], Pm["W1"], Pm["b1"], Pm["W2"], Pm["b2"], bf16=bool(args.bf16), rows16=rows16)
    x = torch.Tensor(system.get_positions()).to(dev)
    w = torch.randn(NN, 3, device=dev)
    d, uhat, dd, ddel = ops.edge_geom(x, topo, w)
    h, hd, mb, mdb = [torch.randn(NN, F_, device=dev) for _ in range(4)]
    if fn.rows16:                                   # (the kernels of this run gather bf16 mirrors)
        h, hd, mb, mdb = [ops.rows_to_bf16(v) for v in (h, hd, mb, mdb)]
    d_b, dd_b = torch.zeros(E, device=dev), torch.zeros(E, device=dev)

    def timed(fn_, reps=10):
        fn_()
        e0, e1 = _events(2)
        e0.record()
        for _ in range(reps):
            fn_()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    kb_ms = timed(lambda: ops.cfconv_bwd(fn, d, dd, topo, h, hd, mb, mdb, d_b, dd_b, True))
    k_ms = timed(lambda: ops.cfconv_fwd(fn, d, dd, h, hd, topo))
    tiles = int(((topo.ell.cnt + 15) // 16).sum())
    GP, FT = (32 if G_ <= 32 else 64), (4 if F_ <= 64 else 8)
    NT_, KS_ = GP // 16, GP // 4
    # cfconv_bwd<DUAL, THETA>, f32 MFMA per 16-edge tile: layer-1 recompute 2 NT KS, gW2 2 x 4 FT NT, s_db / s_b 2 x 4 FT NT,
    # g_db / g_b 2 NT KS, gW1 2 x 4 NT NT  (352 at GP = 32, FT = 8)
    mfma_bwd = 2 * NT_ * KS_ + 8 * FT * NT_ + 8 * FT * NT_ + 2 * NT_ * KS_ + 8 * NT_ * NT_
    etiles = (E + 15) // 16
    exec_bwd = etiles * mfma_bwd * 2048.0
    if args.bf16:                                   # v_mfma_f32_16x16x32_bf16: K = 32 per instruction, 16384 flop
        mfma_per_tile = 2 * (GP // 32) * (GP // 16) + 2 * (GP // 32) * FT
        executed, peak, insn = tiles * mfma_per_tile * 16384.0, MFMA_BF16_PEAK_TF, "v_mfma_f32_16x16x32_bf16 x 16384"
        kname = "cfconv_fwd_bf16_kernel<32,8,true>"
    else:                                           # v_mfma_f32_16x16x4_f32: 2048 flop
        mfma_per_tile = 2 * (GP // 4) * (GP // 16) + 2 * (GP // 4) * FT    # primal + tangent, both Dense layers
        executed, peak, insn = tiles * mfma_per_tile * 2048.0, MFMA_F32_PEAK_TF, "v_mfma_f32_16x16x4_f32 x 2048"
        kname = "cfconv_fwd_kernel<32,8,true>"
    useful = 2.0 * (2 * E) * 2.0 * G_ * (G_ + F_)                            # directed slots x (primal + tangent)
    step_flops = 21.0 * schnet_flops_forward(N, E / R, A_, F_, G_, NC) * R * (T - 1)
    # mixed roof of the whole step: the filter network's products run on the operand type of the run (bf16:
    # 2.5 PF dense; f32: 157.3 TF), every other product (node-level Dense layers, gather-multiply-sum) is f32
    filt_flops = 21.0 * NC * 2.0 * (E / R) * G_ * (G_ + F_) * R * (T - 1)
    filt_peak = MFMA_BF16_PEAK_TF if args.bf16 else MFMA_F32_PEAK_TF
    t_min = filt_flops / (filt_peak * 1e12) + (step_flops - filt_flops) / (MFMA_F32_PEAK_TF * 1e12)
    std = R == 8 and bool(args.bf16)
    bname = "cfconv_bwd_bf16_kernel<32, 8, true, true, false>" if args.bf16 else "cfconv_bwd_kernel<32, 8, true, true>"
    if rows16:
        bname = "cfconv_bwd_bf16_kernel<32, 8, true, true, true>"
    cnt, why = _counters("schnet4096rows" if rows16 else "schnet4096", bname) if std else (None, "other geometry")
    bpeak = MFMA_BF16_PEAK_TF if args.bf16 else MFMA_F32_PEAK_TF
    out["roofline"] = {
        "bound": "mfma", "kernel": bname.replace(", ", ",") + " (reverse sweep of the filter network with parameter gradients: "
                                   "with the forward + tangent sweep the largest share of the pass)",
        "achieved": exec_bwd / (kb_ms * 1e-3) / 1e12, "peak": bpeak, "unit": "TFLOP/s",
        "frac": exec_bwd / (kb_ms * 1e-3) / 1e12 / bpeak, "kernel_ms": kb_ms,
        "frac_of_f32_mfma_peak": exec_bwd / (kb_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF,
        "traffic": cnt.get("hbm_bytes_per_launch") if cnt else None,
        "share_of_pass": cnt.get("share") if cnt else None, "mfma_busy": cnt.get("mfma_busy") if cnt else None,
        "valu_busy": cnt.get("valu_busy") if cnt else None, "wait_frac": cnt.get("wait_frac") if cnt else None,
        "counters": ("profiles/pmc_schnet4096.json: %s, %.1f us under rocprofv3" % (cnt["name"], cnt["avg_us"])) if cnt else why,
        "forward_kernel": {"kernel": kname + " (filter MLP + gather-multiply-sum, primal + tangent)", "kernel_ms": k_ms,
                           "executed_tflops": executed / (k_ms * 1e-3) / 1e12, "peak": peak,
                           "frac": executed / (k_ms * 1e-3) / 1e12 / peak, "useful_tflops": useful / (k_ms * 1e-3) / 1e12},
        "step_roof": {"frac": t_min / (el / steps), "t_min_ms": t_min * 1e3, "t_pass_ms": el / steps * 1e3,
                      "step_tflops": step_flops / (el / steps) / 1e12, "filter_share_of_flops": filt_flops / step_flops,
                      "model": "SURVEY 8d flops per MD step (21 x forward): filter-network products / %s + all other products / "
                               "157.3 TF (f32 MFMA) = the least time the pass could take; frac = that / measured" % (
                                   "2.5 PF (bf16 MFMA operands)" if args.bf16 else "157.3 TF (f32 MFMA)")},
        "note": "achieved = %d 16-edge tiles x %d x 2048 flop -- the arithmetic of the f32 kernel's v_mfma_f32_16x16x4_f32 count; the "
                "bf16 kernel does the same products in 80 wider instructions, and its fraction of the 2.5 PF bf16 peak is small "
                "by construction: the sweep is bound by its gathers (4-8 node rows per edge) and fp32 VALU work, see mfma_busy / "
                "valu_busy / wait_frac (the filter network is recomputed from d: "
                "nothing edge-sized was saved) -- over the kernel time (HIP events); counters of the same kernel from "
                "profiles/pmc_schnet4096.json (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz), FETCH / WRITE).  "
                "forward_kernel: %d 16-slot tiles x %d %s flop (G padded to %d; every undirected edge is evaluated from both "
                "ends), E = %d edges%s.  step_roof prices the whole pass "
                "against the mixed roof of its operand types" % (
                    etiles, mfma_bwd, tiles, mfma_per_tile, insn, GP, E,
                    "; with bf16 operands its Dense layers shrink to 20 MFMAs per tile and it is bound by its f32 VALU work, so "
                    "its fraction of the 2.5 PF bf16 peak is small by construction" if args.bf16 else "")}
    if with_cpu and world == 1:               # (not under the profiler: tools/prof_round3.sh passes --no-cpu-baseline)
        try:
            out["config"]["single_system"] = schnet_single_system(dev, bool(args.bf16), rows16=rows16)
        except Exception as e:
            out["config"]["single_system"] = {"error": "%s: %s" % (type(e).__name__, e)}

    def cpu_part():
        small = cpu_baseline_schnet()
        try:
            # THE TIMED GEOMETRY (8 x 4096 beads, 459 k edges: the eager pass on stored lists, many-row chains, 65 536-slot
            # grids): first and last replica against the oracle for 2 steps; the oracle's wall time on those two runs is the
            # like-for-like CPU figure (one 4096-bead replica at a time, as the reference's sim_list loop would run them)
            par = parity_schnet_stacked(dev, bool(args.bf16), R=8, size=8, T=3, stride=2, sample=(0, 7), seed=2000, rows16=rows16)
            out["cpu_baseline"] = {
                "value": 1.0 / par["oracle_s_per_md_step"], "unit": "MD steps/s", "cores": par["oracle_threads"],
                "threads": par["oracle_threads"], "host_cores": par["host_cores"], "kind": "port",
                "protocol": "wall time of the two oracle trajectories of parity_sampled (no repeats: ~10 s per MD step)",
                "sample": "2 trajectories x 2 steps (fwd + rdf loss + adjoint) of ONE 4096-bead replica of the timed workload, "
                          "oracle/ on %d torch threads" % par["oracle_threads"],
                "parity_sampled": par, "small_box": small}
        except Exception as e:
            out["cpu_baseline"] = dict(small, parity_sampled={"error": "%s: %s" % (type(e).__name__, e)})
        try:
            out["cpu_baseline"]["parity_small_boxes"] = parity_schnet_stacked(dev, bool(args.bf16), rows16=rows16)
        except Exception as e:
            out["cpu_baseline"]["parity_small_boxes"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if with_cpu and world == 1:
        _later(cpu_part)
    return out


# ====================================================================================== config #3: 192-atom water, SchNet
def _golden_water192():
    """tests/golden/gnn_traj_water192.npz (G14): the 64-molecule water box of the reference's data/water_init_64.xyz, the
    reference's own SchNet(A128, F128, G32, 3 conv) weights from torch.manual_seed(0), velocities, and what the REFERENCE
    computed from them on CPU (8 NH-Verlet steps, O-H g(r), parameter gradients) -- data, generated by
    tests/golden/make_goldens.py in the build container."""
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "gnn_traj_water192.npz"), allow_pickle=False))
    sd = {k[4:]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("sd__")}
    prm = {"n_atom_basis": int(g["n_atom_basis"]), "n_filters": int(g["n_filters"]), "n_gaussians": int(g["n_gaussians"]),
           "n_convolutions": int(g["n_convolutions"]), "cutoff": float(g["cutoff"])}
    return g, sd, prm


def build_water192(dev, R=1):
    """R = 1: the golden's box.  R > 1: R copies of it stacked into one System (System.replicate: replicas never interact,
    one thermostat chain each) -- what the reference's sim_list loop (demo/fit_rdf_gnn.py:386-399) runs one after the other."""
    from mdgrad_amd import potentials as P
    from mdgrad_amd.interface import GNNPotentials, PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    from mdgrad_amd.nn import get_model
    from mdgrad_amd.system import System
    g, sd, prm = _golden_water192()
    system = System(positions=np.asarray(g["pos"], dtype=np.float64), cell=np.asarray(g["cell"], dtype=np.float64),
                    numbers=g["numbers"], masses=np.asarray(g["masses"], dtype=np.float64), device=dev)
    system.set_velocities(np.asarray(g["vel"], dtype=np.float64))
    if R > 1:
        system = system.replicate(R)
    net = get_model(prm)
    net.load_state_dict(sd)
    gnn = GNNPotentials(system, net, cutoff=float(g["cutoff"]))
    prior = PairPotentials(system, P.ExcludedVolume(float(g["prior_sigma"]), float(g["prior_epsilon"]), 12), cutoff=float(g["cutoff"]))
    integ = NoseHooverChain(Stack({"gnn": gnn, "prior": prior}), system, T=float(g["T"]), num_chains=int(g["chains"]),
                            Q=float(g["Q"]), adjoint=True).to(dev)
    return g, sd, prm, system, net, gnn, integ


def run_water192(args, rank, world, dev, mdist, with_cpu=True, steps=None, warmup=None, bf16=False):
    """BASELINE config #3 / SURVEY 8d M3: SchNet A128/F128/G32/3 conv + ExcludedVolume prior on the 64-molecule water box,
    cutoff 5, NoseHooverChain, dt = 0.25 fs (the golden's), 20 steps forward + O-H RDF loss + analytic adjoint + Adam, one
    system per GPU (HIP-graph replay of the per-step launches), f32."""
    from mdgrad_amd.observable import rdf
    from mdgrad_amd.sovlers import odeint_adjoint
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    g, sd, prm, system, net, gnn, integ = build_water192(dev)
    net.filter_bf16 = bool(bf16)                          # (bf16 MFMA operands in the filter network: a reported variant)
    T = 21 if args.frames is None or args.workload != "water192" else args.frames
    dt = float(g["dt"])
    obs = rdf(system, nbins=40, r_range=(0.6, 5.0), index_tuple=(g["idx_O"].tolist(), g["idx_H"].tolist()))
    target = torch.ones(40, device=dev)
    params = list(integ.parameters())
    opt = torch.optim.Adam(params, lr=1e-6)
    y0_dev = tuple(x.clone() for x in integ.get_inital_states(wrap=True))
    # ---- parity against the REFERENCE's own output (before any optimizer step): the golden's 8 steps
    par = None
    if rank == 0:
        nf = g["q_t"].shape[0]
        t9 = torch.Tensor([dt * i for i in range(nf)]).to(dev)
        v_t, q_t, pv_t = odeint_adjoint(integ, tuple(x.clone() for x in y0_dev), t9, method="NH_verlet")
        gr = obs(q_t[::2])[2]
        loss = gr.pow(2).mean() + q_t[-1].pow(2).mean() * 1e-3 + v_t[-1].pow(2).sum() * 1e-2
        loss.backward()
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params]).cpu().numpy()
        ref = g["grad_flat"]
        par = {"vs": "the reference's CPU run (golden G14, tests/golden/gnn_traj_water192.npz)", "steps": int(nf - 1),
               "max_abs_dq": float(np.abs(q_t.detach().cpu().numpy() - g["q_t"]).max()),
               "max_abs_dg": float(np.abs(gr.detach().cpu().numpy() - g["g"]).max()),
               "rel_dtheta": float(np.abs(flat - ref).max() / np.abs(ref).max()),
               "cos_dtheta": float((flat.astype(np.float64) * ref).sum() / (np.linalg.norm(flat.astype(np.float64)) * np.linalg.norm(ref)))}
        opt.zero_grad(set_to_none=True)
    t = torch.Tensor([dt * i for i in range(T)]).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        y0 = tuple(x.clone() for x in y0_dev)
        v_t, q_t, pv_t = odeint_adjoint(integ, y0, t, method="NH_verlet")
        loss = (obs(q_t[::2])[2] - target).pow(2).mean()
        loss.backward()
        mdist.all_reduce_grads(params)
        opt.step()
        return loss, q_t

    for _ in range(warmup):
        step()
    mdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, q_last = step()
    torch.cuda.synchronize()
    mdist.barrier()
    el_rank = time.perf_counter() - t0
    el = mdist.max_over_ranks(el_rank, dev)
    if not (_finite(q_last) and all(_finite(p) for p in params)):
        raise SystemExit("bench: non-finite trajectory or parameters -- the measurement would be invalid")
    N = len(system)
    md_steps = (T - 1) * world * steps
    topo = gnn.inputs["_topo"]
    E = int(topo.n_edges)
    A_, F_, G_, NC = prm["n_atom_basis"], prm["n_filters"], prm["n_gaussians"], prm["n_convolutions"]
    step_flops = 21.0 * schnet_flops_forward(N, E, A_, F_, G_, NC)
    sec_per_step = el / (steps * (T - 1))
    out = {"metric": "MD steps/sec (fwd+adjoint), 192-atom water SchNet NHC (BASELINE config #3)", "value": md_steps / el,
           "unit": "MD steps/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": el / steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "bf16 filter MFMA operands, f32 accumulate" if bf16 else F32_SCHNET, "data": "synthetic",
           "config": {"workload": "64-molecule water box (192 atoms, golden G14 geometry), SchNet A%d F%d G%d %d conv + "
                                  "ExcludedVolume prior, cutoff 5, NoseHooverChain(Q=50, 5 chains), %d steps fwd + O-H RDF(40 "
                                  "bins) loss + analytic adjoint + Adam; one system per GPU, HIP-graph replay" % (A_, F_, G_, NC, T - 1),
                      "replicas_per_gpu": 1, "parallelism": "replica-dp%d" % world, "loss": float(loss.detach()), "edges": E,
                      "us_per_md_step": sec_per_step * 1e6}}
    out["config"]["dist"] = _dist_record(mdist, dev, params, el_rank / steps * 1e3)
    if rank != 0:
        return out
    out["config"]["parity_reference_golden"] = par
    out["roofline"] = {"bound": "mfma", "kernel": "whole MD step (launch-bound: one 192-atom system, E = %d edges)" % E,
                       "achieved": step_flops / sec_per_step / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                       "frac": step_flops / sec_per_step / 1e12 / MFMA_F32_PEAK_TF, "traffic": None,
                       "note": "SURVEY 8d: 21 x the flops of one forward energy evaluation per MD step (1 force + 2 force-vjp "
                               "evaluations) over the measured time per step; at 192 atoms the step is a chain of ~70 graph "
                               "nodes of 5-15 us each: bound by launch latency, not by the matrix cores"}
    if with_cpu and world == 1:
        def cpu_part():
            import oracle as O
            total, usable = _host_cpus()
            threads = max(1, min(usable, 16))
            torch.set_num_threads(threads)
            cell = torch.from_numpy(np.asarray(g["cell"], dtype=np.float32))
            nf = g["q_t"].shape[0]
            tt = torch.Tensor([dt * i for i in range(nf)])

            def one():
                gnn_o = O.SchNetTerm(sd, g["numbers"], float(g["cutoff"]), cell)
                prior_o = O.PairTerm("lj", torch.tensor([float(g["prior_sigma"]), float(g["prior_epsilon"])]), float(g["cutoff"]),
                                     cell, p=12, q=0, c=0)
                eom = O.NHCOracle(O.ModelOracle([gnn_o, prior_o]), torch.from_numpy(g["masses"]), float(g["T"]), float(g["Q"]),
                                  int(g["chains"]))
                traj = O.odeint_oracle(eom, (torch.from_numpy(g["vel"]), torch.from_numpy(g["pos"]), torch.zeros(5)), tt)
                leaves = [x.clone().requires_grad_(True) for x in traj]
                gr = O.rdf_oracle(leaves[1][::2], cell, 40, (0.6, 5.0), (g["idx_O"].tolist(), g["idx_H"].tolist()))[2]
                (gr - 1).pow(2).mean().backward()
                O.adjoint_oracle(eom, traj, [x.grad if x.grad is not None else torch.zeros_like(x) for x in leaves], tt)
            one()
            ts = []
            for _ in range(3):
                t0_ = time.perf_counter()
                one()
                ts.append(time.perf_counter() - t0_)
            med = sorted(ts)[1]
            out["cpu_baseline"] = {"value": (nf - 1) / med, "unit": "MD steps/s", "cores": threads, "threads": threads,
                                   "host_cores": total, "usable_cores": usable, "kind": "port",
                                   "sample": "1 warm-up + median of 3 runs of %d steps (fwd + O-H rdf loss + adjoint) of the timed "
                                             "system itself, oracle/ (autograd double backward like the reference) on %d torch "
                                             "threads; median %.2f s" % (nf - 1, threads, med)}
        _later(cpu_part)
    return out



def run_water192_stacked(args, rank, world, dev, mdist, R=64, steps=10, warmup=3):
    """config #3's system the many-replica way -- R = 64 copies of the 192-atom water box stacked into ONE
    trajectory (12 288 atoms; per-replica thermostats and neighbour lists), the way the 108-atom LJ headline runs 16 384
    replicas: one system per GPU leaves the chip idle between 5-15 us launches, R systems fill the same launches.  Parity:
    replicas 0 and R - 1 start from the golden's state, so the first 8 steps of each must be the REFERENCE's own run (G14)
    and the summed parameter gradient R x the reference's."""
    from mdgrad_amd.observable import rdf
    from mdgrad_amd.sovlers import odeint_adjoint
    g, sd, prm, system, net, gnn, integ = build_water192(dev, R)
    base = build_water192(dev, 1)[3]
    N = len(base)
    T = 21
    dt = float(g["dt"])
    obs = rdf(base, nbins=40, r_range=(0.6, 5.0), index_tuple=(g["idx_O"].tolist(), g["idx_H"].tolist()))
    target = torch.ones(40, device=dev)
    params = list(integ.parameters())
    opt = torch.optim.Adam(params, lr=1e-6)
    y0_dev = tuple(x.clone() for x in integ.get_inital_states(wrap=True))
    par = None
    if rank == 0:
        nf = g["q_t"].shape[0]
        t9 = torch.Tensor([dt * i for i in range(nf)]).to(dev)
        v_t, q_t, pv_t = odeint_adjoint(integ, tuple(x.clone() for x in y0_dev), t9, method="NH_verlet")
        qr, vr = q_t.reshape(nf, R, N, 3), v_t.reshape(nf, R, N, 3)
        loss = 0.0
        for r in range(R):                                    # the golden's loss, per replica
            gr = obs(qr[::2, r])[2]
            loss = loss + gr.pow(2).mean() + qr[-1, r].pow(2).mean() * 1e-3 + vr[-1, r].pow(2).sum() * 1e-2
        loss.backward()
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params]).cpu().numpy() / R
        ref = g["grad_flat"]
        qn = qr.detach().cpu().numpy()
        par = {"vs": "the reference's CPU run (golden G14) on replicas 0 and %d of the stack" % (R - 1), "steps": int(nf - 1),
               "max_abs_dq": float(max(np.abs(qn[:, 0] - g["q_t"]).max(), np.abs(qn[:, R - 1] - g["q_t"]).max())),
               "max_abs_dq_any_replica": float(np.abs(qn - g["q_t"][:, None]).max()),
               "rel_dtheta": float(np.abs(flat - ref).max() / np.abs(ref).max()),
               "cos_dtheta": float((flat.astype(np.float64) * ref).sum() / (np.linalg.norm(flat.astype(np.float64)) * np.linalg.norm(ref)))}
        opt.zero_grad(set_to_none=True)
    # timed passes: every replica its own velocities (a scaled copy of the golden's: 0.9 .. 1.1)
    v0 = y0_dev[0].reshape(R, N, 3) * torch.linspace(0.9, 1.1, R, device=dev)[:, None, None]
    y0_dev = (v0.reshape(-1, 3).contiguous(),) + tuple(y0_dev[1:])
    t = torch.Tensor([dt * i for i in range(T)]).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        y0 = tuple(x.clone() for x in y0_dev)
        v_t, q_t, pv_t = odeint_adjoint(integ, y0, t, method="NH_verlet")
        loss = (obs(q_t[::2].reshape(-1, N, 3))[2] - target).pow(2).mean()       # (replicas as further frames of the same box)
        loss.backward()
        mdist.all_reduce_grads(params)
        opt.step()
        return loss, q_t

    for _ in range(warmup):
        step()
    mdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, q_last = step()
    torch.cuda.synchronize()
    mdist.barrier()
    el_rank = time.perf_counter() - t0
    el = mdist.max_over_ranks(el_rank, dev)
    if not (_finite(q_last) and all(_finite(p) for p in params)):
        raise SystemExit("bench: non-finite trajectory or parameters -- the measurement would be invalid")
    md_steps = R * (T - 1) * world * steps
    E = int(gnn.inputs["_topo"].n_edges)
    A_, F_, G_, NC = prm["n_atom_basis"], prm["n_filters"], prm["n_gaussians"], prm["n_convolutions"]
    step_flops = 21.0 * schnet_flops_forward(N * R, E, A_, F_, G_, NC)
    sec_per_step = el / (steps * (T - 1))
    out = {"metric": "MD steps/sec (fwd+adjoint), 192-atom water SchNet NHC, %d stacked replicas" % R, "value": md_steps / el,
           "unit": "MD steps/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": el / steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": F32_SCHNET, "data": "synthetic",
           "config": {"workload": "%d stacked 64-molecule water boxes (192 atoms each), SchNet A%d F%d G%d %d conv + prior, %d steps "
                                  "fwd + O-H RDF loss + adjoint + Adam" % (R, A_, F_, G_, NC, T - 1),
                      "replicas_per_gpu": R, "parallelism": "replica-dp%d" % world, "loss": float(loss.detach()), "edges": E,
                      "us_per_stacked_md_step": sec_per_step * 1e6}}
    out["config"]["dist"] = _dist_record(mdist, dev, params, el_rank / steps * 1e3)
    if rank == 0:
        out["config"]["parity_reference_golden"] = par
        out["roofline"] = {"bound": "mfma", "kernel": "whole MD step of the stack (%d atoms, E = %d edges)" % (N * R, E),
                           "achieved": step_flops / sec_per_step / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                           "frac": step_flops / sec_per_step / 1e12 / MFMA_F32_PEAK_TF, "traffic": None}
    return out


# ====================================================================================== 4096-atom LJ liquid
def lj_liquid(n_side, rho, rng, jitter=0.05):
    L = (n_side ** 3 / rho) ** (1 / 3)
    g = np.stack(np.meshgrid(*[np.arange(n_side)] * 3, indexing="ij"), -1).reshape(-1, 3) * (L / n_side)
    return np.mod(g + rng.uniform(-jitter, jitter, g.shape) * (L / n_side), L), L


def cpu_baseline_lj4096(budget_s=10.0):
    import oracle as O
    rng = np.random.default_rng(5)
    pos, L = lj_liquid(10, 0.845, rng)
    vel = rng.normal(0, 1.0, pos.shape)
    cell = torch.tensor([L] * 3, dtype=torch.float32)
    nsteps = 2
    t = torch.Tensor([0.005 * i for i in range(nsteps + 1)])

    def one():
        term = O.PairTerm("lj", torch.tensor([1.0, 1.0]), 2.5, cell, p=12, q=6, c=1)
        eom = O.NHCOracle(O.ModelOracle([term]), torch.full((len(pos),), 1.008), 1.0, 50.0, 5)
        traj = O.odeint_oracle(eom, (torch.Tensor(vel), torch.Tensor(pos), torch.zeros(5)), t)
        leaves = [x.clone().requires_grad_(True) for x in traj]
        leaves[1].pow(2).mean().backward()
        O.adjoint_oracle(eom, traj, [x.grad if x.grad is not None else torch.zeros_like(x) for x in leaves], t)

    out = _cpu_timed(one, nsteps, "a 1000-atom LJ liquid (fwd + adjoint; dense N^2 neighbour search of the reference algorithm)")
    out["sample"] += ("; NOT the timed geometry (64 x 4096 atoms): BASELINE.md has the reference at 0.34 steps/s for 4000 atoms")
    return out


def parity_lj_large(dev, R=64, n_side=10, T=11):
    """The kernels the lj4096 leg times (csrc/traj_large.hip: 64 stacked replicas per launch, Verlet reuse with device-side
    rebuild decisions, cell-sweep RDF) on a system the oracle finishes in seconds: 64 x 1 000 atoms, 10 steps + RDF loss +
    adjoint, first / middle / last replica of the launch against oracle/ on the same inputs."""
    import oracle as O
    from mdgrad_amd import ops
    from mdgrad_amd import potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    from mdgrad_amd.observable import rdf
    from mdgrad_amd.system import System, Atoms
    rng = np.random.default_rng(77)
    pos1, L = lj_liquid(n_side, 0.845, rng)
    N = len(pos1)
    system = System(Atoms(positions=pos1, cell=[L, L, L], numbers=np.ones(N)), device=dev)
    mdl = P.LennardJones(1.0, 1.0)
    integ = NoseHooverChain(Stack({"pair": PairPotentials(system, mdl, cutoff=2.5)}), system, T=1.0, num_chains=5,
                            Q=50.0).to(dev)
    integ.fused_large = True
    integ.fuse_observables = False
    spec = integ.fused_spec("NH_verlet")
    assert spec is not None and spec.large
    pos = np.stack([lj_liquid(n_side, 0.845, rng)[0] for _ in range(R)]).astype(np.float32)
    vel = rng.normal(0, 1.0, (R, N, 3)).astype(np.float32)
    t = torch.Tensor([0.005 * i for i in range(T)])
    obs = rdf(system, nbins=100, r_range=(0.75, 2.5))
    sample = sorted({0, R // 2, R - 1})
    was, ops.RDF_LIST_ATOMS = ops.RDF_LIST_ATOMS, 512          # the cell-sweep RDF of the timed leg (normally from 2 048 atoms)
    try:
        return _parity_lj_large_body(dev, O, ops, obs, spec, mdl, system, pos, vel, t, sample, R, N, L, T)
    finally:
        ops.RDF_LIST_ATOMS = was


def _parity_lj_large_body(dev, O, ops, obs, spec, mdl, system, pos, vel, t, sample, R, N, L, T, stride=5):
    total, usable = _host_cpus()
    threads = max(1, min(usable, 32))
    torch.set_num_threads(threads)
    t_or = 0.0
    v_t, q_t, pv_t = ops.fused_traj(torch.from_numpy(vel).to(dev), torch.from_numpy(pos).to(dev),
                                    torch.zeros(R, 5, device=dev), t.to(dev), spec.flat_params(), spec)
    cell = torch.tensor([L] * 3, dtype=torch.float32)
    mass = torch.full((N,), float(system.get_masses()[0]))
    dq = dg = dth = 0.0
    theta_now = torch.tensor([float(mdl.sigma.detach()), float(mdl.epsilon.detach())])   # (the optimizer may have moved them)
    for r in sample:
        mdl.zero_grad()
        _, _, g = obs(q_t[r, ::stride])
        (g - 1).pow(2).mean().backward(retain_graph=True)
        gth_hip = torch.stack([mdl.sigma.grad.reshape(()), mdl.epsilon.grad.reshape(())]).cpu()
        t0 = time.perf_counter()
        term = O.PairTerm("lj", theta_now, 2.5, cell, p=12, q=6, c=1)
        eom = O.NHCOracle(O.ModelOracle([term]), mass, 1.0, 50.0, 5)
        traj = O.odeint_oracle(eom, (torch.from_numpy(vel[r]), torch.from_numpy(pos[r]), torch.zeros(5)), t)
        leaves = [x.clone().requires_grad_(True) for x in traj]
        _, _, go = O.rdf_oracle(leaves[1][::stride], cell, 100, (0.75, 2.5))
        (go - 1).pow(2).mean().backward()
        _, gth = O.adjoint_oracle(eom, traj, [x.grad if x.grad is not None else torch.zeros_like(x) for x in leaves], t)
        t_or += time.perf_counter() - t0
        dq = max(dq, float((q_t[r].detach().cpu() - traj[1]).abs().max()))
        dg = max(dg, float((g.detach().cpu() - go.detach()).abs().max()))
        dth = max(dth, float(((gth_hip - gth).abs() / gth.abs().max()).max()))
    return {"replicas": list(sample), "atoms_per_replica": N, "stacked_replicas": R, "steps": T - 1,
            "oracle_s_per_md_step": t_or / (len(sample) * (T - 1)), "oracle_threads": threads, "host_cores": total,
            "max_abs_dq": dq, "max_abs_dg": dg, "rel_dtheta": dth,
            "note": "%d stacked replicas x %d atoms on the kernels of the timed launch (multi-launch path, stored candidate "
                    "lists reused across steps, cell-sweep RDF on every %d. frame), %d steps + RDF loss + adjoint: replicas %s "
                    "vs oracle/ on the same inputs, worst of them (the 64 x 4096 geometry is also pinned in "
                    "tests/test_gpu_secondary_pins.py)" % (R, N, stride, T - 1, list(sample))}


def run_lj4096(args, rank, world, dev, mdist, with_cpu=True, steps=None, warmup=None):
    from mdgrad_amd import ops
    from mdgrad_amd import potentials as P
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    from mdgrad_amd.observable import rdf
    from mdgrad_amd.system import System, Atoms
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    R = 64 if args.replicas is None or args.workload != "lj4096" else args.replicas
    T = 51 if args.frames is None or args.workload != "lj4096" else args.frames
    rng = np.random.default_rng(3000 + rank)
    pos1, L = lj_liquid(16, 0.845, rng)
    N = len(pos1)
    system = System(Atoms(positions=pos1, cell=[L, L, L], numbers=np.ones(N)), device=dev)
    mdl = P.LennardJones(1.0, 1.0)
    integ = NoseHooverChain(Stack({"pair": PairPotentials(system, mdl, cutoff=2.5)}), system, T=1.0, num_chains=5,
                            Q=50.0).to(dev)
    spec = integ.fused_spec("NH_verlet")
    assert spec is not None and spec.large
    pos = torch.from_numpy(np.stack([lj_liquid(16, 0.845, rng)[0] for _ in range(R)]).astype(np.float32)).to(dev)
    vel = torch.from_numpy(rng.normal(0, 1.0, (R, N, 3)).astype(np.float32)).to(dev)
    pv0 = torch.zeros(R, 5, device=dev)
    obs = rdf(system, nbins=100, r_range=(0.75, 2.5))
    target = torch.ones(100, device=dev)
    t = torch.Tensor([0.005 * i for i in range(T)]).to(dev)
    params = list(integ.parameters())
    opt = torch.optim.Adam(params, lr=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        v_t, q_t, pv_t = ops.fused_traj(vel, pos, pv0, t, spec.flat_params(), spec)
        loss = (obs(q_t[:, ::5])[2] - target).pow(2).mean()
        loss.backward()
        mdist.all_reduce_grads(params)
        opt.step()
        return loss, q_t

    for _ in range(warmup):
        step()
    mdist.barrier()
    torch.cuda.synchronize()
    tr = _PassTrace("lj4096")
    t0 = time.perf_counter()
    tr.start()
    for _ in range(steps):
        loss, q_last = step()
        tr.tick()
    torch.cuda.synchronize()
    mdist.barrier()
    el_rank = time.perf_counter() - t0
    tr.done()
    el = mdist.max_over_ranks(el_rank, dev)
    if not (_finite(q_last) and all(_finite(p) for p in params)):
        raise SystemExit("bench: non-finite trajectory or parameters -- the measurement would be invalid")
    md_steps = R * (T - 1) * world * steps
    out = {"metric": "MD steps/sec (fwd+adjoint), 4096-atom LJ liquid NHC", "value": md_steps / el, "unit": "MD steps/s",
           "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": el / steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "LJ(1,1) liquid, 4096 atoms, rho 0.845, cutoff 2.5, NoseHooverChain(Q=50, 5 chains), %d "
                                  "steps fwd + RDF(100 bins, every 5th frame) loss + adjoint; %d replicas/GPU per pass, "
                                  "cell-binned neighbour search fused into every force evaluation" % (T - 1, R),
                      "replicas_per_gpu": R, "parallelism": "replica-dp%d" % world, "loss": float(loss.detach())}}
    out["config"]["dist"] = _dist_record(mdist, dev, params, el_rank / steps * 1e3)
    if rank != 0:
        return out
    ell = ops.build_ell(pos[0], spec.cell_struct, 2.5)
    Pn = int(ell.half_list()[0].shape[0])
    bytes_step = 60.0 * Pn + 316.0 * N                               # SURVEY 8d B_step
    sec_per_step = el / md_steps * world
    # per MD step + adjoint interval: one force sweep (40 flop per directed pair) and two force + Hessian.w sweeps (70)
    useful = (40.0 + 2.0 * FLOP_PER_PAIR_ADJ) * 2.0 * Pn / sec_per_step
    # counters of the same workload (profiles/pmc_lj4096.json, 64 replicas x 51 frames): HBM bytes of a whole pass summed over
    # its kernels, and the issue-side picture of the kernel with the largest share of the pass
    std = R == 64 and T == 51
    dom, why = _counters("lj4096", "large_adj_") if std else (None, "other geometry")      # (large_adj_tiled / large_adj_listed)
    # the launches of a trajectory are issued for groups of replicas on concurrent streams (csrc/traj_large.hip lg_group_count)
    groups = int(os.environ.get("MDG_LARGE_STREAMS", "0")) or (3 if R >= 48 else (2 if R >= 8 else 1))
    pj = _profile_json("pmc_lj4096.json") if std else None
    traffic = _pass_traffic("lj4096", 3) if (dom is not None) else None     # (the profiled run: 1 warm-up + 2 timed passes)
    B_H = (12.0 * Pn + 48.0 * N) * R / groups                               # fused force + Hessian.w sweep, per launch
    dominant = None
    if dom is not None:
        sec_k = dom["avg_us"] * 1e-6
        dominant = {"kernel": dom["name"], "share_of_pass": dom["share"], "avg_us_rocprof": dom["avg_us"],
                    "algorithmic_bytes_per_launch": B_H, "algorithmic_gbs": B_H / sec_k / 1e9,
                    "algorithmic_frac_of_hbm_peak": B_H / sec_k / 1e9 / HBM_PEAK_GBS,
                    "hbm_bytes_per_launch_measured": dom.get("hbm_bytes_per_launch"), "hbm_gbs_measured": dom.get("hbm_gbs"),
                    "valu_busy": dom.get("valu_busy"), "wait_frac": dom.get("wait_frac"), "vgpr": dom.get("vgpr"),
                    "replica_groups_on_concurrent_streams": groups,
                    "binding": "VALU issue (0.70 when the launch runs alone, profiles/r05c_lj4096_kernel_stats.txt; under the "
                               "profiler the groups' launches overlap and each sees a share of the chip): SQ_ACTIVE_INST_VALU x 4 / "
                               "(1024 SIMDs x duration x 2.4 GHz) = %.2f; its measured HBM traffic is %.2f x the algorithmic bytes "
                               "at %.0f GB/s" % (
                                   dom.get("valu_busy", float("nan")), dom.get("hbm_bytes_per_launch", float("nan")) / B_H,
                                   dom.get("hbm_gbs", float("nan")))}
    out["roofline"] = {"bound": "hbm", "kernel": "whole MD step (large_prep / large_search_rows / large_fwd_tiled / large_adj_tiled "
                                                 "+ cell-sweep RDF)",
                       "achieved": bytes_step / sec_per_step / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": bytes_step / sec_per_step / 1e9 / HBM_PEAK_GBS,
                       "traffic": traffic, "algorithmic_bytes_per_pass": bytes_step * R * (T - 1),
                       "traffic_over_algorithmic": (traffic / (bytes_step * R * (T - 1))) if traffic else None,
                       "busy_over_span_profiled": pj.get("busy_over_span") if pj else None,
                       "dominant_kernel": dominant if dominant is not None else why,
                       "useful_tflops": useful / 1e12, "useful_frac_of_vector_peak": useful / 1e12 / VEC_F32_PEAK_TF,
                       "note": "SURVEY 8d: B_step = 60 P + 316 N bytes per MD step (P = %d half-list pairs) over the measured "
                               "time per MD step of the whole pass.  `traffic` = HBM bytes of one pass from the FETCH_SIZE / "
                               "WRITE_SIZE counters of every kernel of the workload (profiles/pmc_lj4096.json, refused when the "
                               "kernel sources changed).  The pass is not bandwidth-bound: the kernel with the largest share, the "
                               "listed force + Hessian.w sweep of the adjoint, keeps the SIMDs' VALU issue slots busy (see "
                               "dominant_kernel); searches run one step in ~7 (Verlet reuse, device-side decision)" % Pn}
    def cpu_part():
        small = cpu_baseline_lj4096()
        try:
            # THE TIMED GEOMETRY: the last replica of the 64 x 4096-atom launch itself, 2 steps + RDF loss + adjoint, against
            # the oracle; the oracle's wall time on that run is the like-for-like CPU figure
            import oracle as O
            t3 = torch.Tensor([0.005 * i for i in range(3)])
            par = _parity_lj_large_body(dev, O, ops, obs, spec, mdl, system, pos.cpu().numpy(), vel.cpu().numpy(), t3, (R - 1,),
                                        R, N, L, 3, stride=2)
            out["cpu_baseline"] = {
                "value": 1.0 / par["oracle_s_per_md_step"], "unit": "MD steps/s", "cores": par["oracle_threads"],
                "threads": par["oracle_threads"], "host_cores": par["host_cores"], "kind": "port",
                "protocol": "wall time of the oracle trajectory of parity_sampled (no repeats: several s per MD step)",
                "sample": "1 trajectory x 2 steps (fwd + rdf loss + adjoint) of ONE 4096-atom replica of the timed workload, "
                          "oracle/ (dense N^2 neighbour search of the reference algorithm) on %d torch threads" % par["oracle_threads"],
                "parity_sampled": par, "small_box": small}
        except Exception as e:                     # (reported, never hidden: a failed check is part of the record)
            out["cpu_baseline"] = dict(small, parity_sampled={"error": "%s: %s" % (type(e).__name__, e)})
        try:
            out["cpu_baseline"]["parity_small_boxes"] = parity_lj_large(dev)
        except Exception as e:
            out["cpu_baseline"]["parity_small_boxes"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if with_cpu and world == 1:
        _later(cpu_part)
    return out


# ====================================================================================== output
# What the driver keeps of a run is (a) the LAST JSON line, of which it stores the scalar keys of `config` and the scalar
# entries of `roofline` / `cpu_baseline` (strings cut at ~140 characters), and (b) the last ~8 KB of stdout.  So: every
# secondary workload is printed as its own compact line BEFORE the headline, the headline carries the secondaries' figures
# as flat scalars in `config`, and everything long (notes, nested records) goes to the detail file only
# (gpurun_out/bench_full.json, or $MDG_BENCH_DETAIL).
_DROP = {"note", "hbm_algorithmic_model", "protocol", "batch_s", "threads_tried", "counters", "model", "parity_small_boxes",
         "small_box", "cpu_baseline_sample", "binding"}


def _compact(o, depth=0, cut=150):
    if isinstance(o, dict):
        return {k: _compact(v, depth + 1, cut) for k, v in o.items() if k not in _DROP and not (depth >= 3 and isinstance(v, (dict, list)))}
    if isinstance(o, (list, tuple)):
        return [_compact(v, depth + 1, cut) for v in o][:16]
    if isinstance(o, str) and len(o) > cut:
        return o[:cut - 3] + "..."
    if isinstance(o, float):
        return o if depth <= 1 else float("%.7g" % o)     # (top-level figures exact: value = units / time to the last digit)
    return o


def _flat(name, rec):
    """The figures of a secondary workload as flat scalar keys for the headline's `config`."""
    if not isinstance(rec, dict) or "error" in rec:
        return {name + "_error": str((rec or {}).get("error"))[:140]}
    rl, cb = rec.get("roofline") or {}, rec.get("cpu_baseline") or {}
    par = cb.get("parity_sampled") or rec.get("config", {}).get("parity_reference_golden") or {}
    # (the keys VERDICT r4 #1 names, + the parameter-gradient parity; the rest of a leg is in its own line)
    f = {name + "_md_steps_per_s": rec.get("value"), name + "_ms_per_pass": rec.get("ms_per_step"), name + "_dtype": rec.get("dtype"),
         name + "_kernel_frac": rl.get("frac"), name + "_step_roof_frac": (rl.get("step_roof") or {}).get("frac"),
         name + "_cpu_steps_per_s": cb.get("value"), name + "_parity_max_abs_dq": par.get("max_abs_dq"),
         name + "_parity_rel_dtheta": par.get("rel_dtheta")}
    return {k: (float("%.6g" % v) if isinstance(v, float) else v) for k, v in f.items() if v is not None}


def _line(name, rec):
    """One compact JSON line of a secondary workload (< ~1.5 KB)."""
    if not isinstance(rec, dict) or "error" in rec:
        return json.dumps({"workload": name, "error": str((rec or {}).get("error"))[:300]})
    c = rec.get("config", {})
    o = {"workload": name, "metric": rec["metric"], "value": rec["value"], "unit": rec["unit"], "n_gpus": rec["n_gpus"],
         "steps": rec["steps"], "warmup": rec["warmup"], "ms_per_step": rec["ms_per_step"], "dtype": rec["dtype"],
         "replicas_per_gpu": c.get("replicas_per_gpu"), "per_rank_ms": (c.get("dist") or {}).get("per_rank_ms")}
    rl = rec.get("roofline") or {}
    o["roofline"] = {k: rl.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms") if k in rl}
    if "step_roof" in rl:
        o["roofline"]["step_roof_frac"] = rl["step_roof"].get("frac")
    cb = rec.get("cpu_baseline") or {}
    o["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample") if k in cb}
    par = cb.get("parity_sampled") or c.get("parity_reference_golden")
    if par:
        o["parity"] = {k: v for k, v in par.items() if k in ("max_abs_dq", "max_abs_dg", "rel_dtheta", "cos_dtheta", "replicas", "steps", "vs")}
    for k in ("f32", "bf16", "bf16_f32rows", "steps10", "single_system", "neighbour_list", "md_steps_per_s_traj_only_per_gpu"):
        if k in rec:
            o[k] = rec[k]
        elif k in c:
            o[k] = c[k]
    return json.dumps(_compact(o, cut=90))


def _write_detail(out):
    path = os.environ.get("MDG_BENCH_DETAIL") or os.path.join(ROOT, "gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            json.dump(out, fh)
    except OSError:
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="all", choices=["all", "lj108", "exvol108", "schnet4096", "lj4096", "water192", "water192x64"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200,
                    help="timed passes (default: ~4.5 s of GPU time on the headline workload, so that a coarse utilisation "
                         "sampler sees the device busy)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--replicas", type=int, default=None, help="replicas per GPU (default 16384 / 8 / 64)")
    ap.add_argument("--frames", type=int, default=None, help="saved frames T (T-1 MD steps); default 50 / 53 / 51 / 21")
    ap.add_argument("--dt", type=float, default=0.005)
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--bf16", action="store_true", help="schnet4096: bf16 MFMA operands in the filter network")
    ap.add_argument("--bf16-rows", action="store_true",
                    help="schnet4096: --bf16 and bf16 mirrors of the node rows the convolution kernels gather (a further precision "
                         "option, SchNet.node_rows_bf16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--all-legs", action="store_true", help="with --gpus N > 1: every secondary leg and precision variant (default "
                                                             "there: headline + lj4096 + schnet4096 only)")
    ap.add_argument("--full-line", action="store_true", help="print the un-compacted record as the last line (default: compact "
                                                              "line; the full record goes to gpurun_out/bench_full.json)")
    args = ap.parse_args()
    if args.bf16_rows:
        args.bf16 = True

    from mdgrad_amd import dist as mdist
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain `python bench.py --gpus N`: spawn the N ranks ourselves (one per GPU, rendezvous on
        # 127.0.0.1 at a free port); rank 0 of the children prints the ONE JSON line on our stdout
        if os.environ.get("MDG_SINGLE_DEVICE") != "1" and torch.cuda.device_count() < args.gpus:
            raise SystemExit("--gpus %d but only %d HIP device(s) are visible" % (args.gpus, torch.cuda.device_count()))
        sys.exit(mdist.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    rank, world, dev = mdist.init()
    if dev.type != "cuda":
        raise SystemExit("bench.py needs a HIP device (the hot path has no CPU implementation)")
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    cpu = not args.no_cpu_baseline
    lines = []
    if args.workload == "schnet4096":
        out = run_schnet4096(args, rank, world, dev, mdist, cpu)
    elif args.workload == "lj4096":
        out = run_lj4096(args, rank, world, dev, mdist, cpu)
    elif args.workload == "water192":
        out = run_water192(args, rank, world, dev, mdist, cpu)
    elif args.workload == "water192x64":
        out = run_water192_stacked(args, rank, world, dev, mdist, R=args.replicas or 64, steps=args.steps, warmup=args.warmup)
    elif args.workload == "exvol108":
        out = run_lj108(args, rank, world, dev, mdist, cpu, form="exvol", dt=0.01)
    else:
        global _DEFERRED
        if args.workload == "all" and not args.no_secondary:
            _DEFERRED = []                      # (every timed GPU leg first, the CPU legs afterwards: see _DEFERRED)
        out = run_lj108(args, rank, world, dev, mdist, cpu)
        if args.workload == "all" and not args.no_secondary:
            sec = {}
            import copy
            # (>= 1 s of GPU time per secondary workload)
            # BASELINE config #5 names the bf16 cfconv MFMA: the SchNet workload runs with bf16 filter operands (stated
            # tolerance: tests/test_gpu_config5.py) and reports the all-f32 rate beside it
            # Round 6 (VERDICT r5 next #1 iii): "bf16 cfconv" is taken to cover what the convolution READS as well -- THE
            # schnet4096 leg gathers bf16 mirrors of the node rows (SchNet.node_rows_bf16: operands rounded to bf16, every product
            # and sum f32), inside the bf16 tolerances of tests/test_gpu_secondary_pins.py against the oracle; the leg's 52-step
            # deviation from the all-f32 path rides in its record (`bf16_vs_f32`), and the plain bf16-operand and all-f32 rates
            # are reported beside it
            a16 = copy.copy(args)
            a16.bf16 = True
            a16.bf16_rows = True
            # SURVEY 8d M4 names opt_freq = 52 steps per pass (demo/fit_rdf_gnn.py): THE schnet4096 leg runs that horizon
            # (VERDICT r4 weak #6); the 10-step passes of rounds 2-4 are reported beside it (`steps10`)
            a52 = copy.copy(a16)
            a52.frames_override = 53
            a16.frames_override = 11
            # (warm-up passes: the first Adam step builds its state, and one of the first half-dozen passes of a process has
            #  been seen to take ~100 ms longer than the rest (the caching allocator taking a multi-GB block from the driver for
            #  the first time) -- six / eight warm-up passes keep that out of the timed ones)
            legs = (("exvol108", lambda: run_lj108(args, rank, world, dev, mdist, cpu, form="exvol", dt=0.01, steps=20, warmup=3)),
                    ("schnet4096", lambda: run_schnet4096(a52, rank, world, dev, mdist, cpu, steps=8, warmup=3)),
                    ("lj4096", lambda: run_lj4096(args, rank, world, dev, mdist, cpu, steps=50, warmup=8)),
                    ("water192", lambda: run_water192(args, rank, world, dev, mdist, cpu, steps=20, warmup=4)))
            # N > 1 (the driver's 2 / 4 / 8-GPU scaling runs): the headline and the two workloads BASELINE names with a GPU
            # count (configs #4 and #5) only -- no precision variants, no single-GPU legs -- so that an 8-rank launch stays far
            # inside the driver's time limit; --all-legs restores the full set
            lean = world > 1 and not args.all_legs
            if lean:
                legs = tuple(l for l in legs if l[0] in ("schnet4096", "lj4096"))
            for name, fn in legs:
                try:
                    sec[name] = fn()
                except (Exception, SystemExit) as e:        # a secondary workload must not take the headline down
                    sec[name] = {"error": "%s: %s" % (type(e).__name__, e)}
            if "error" not in sec["schnet4096"] and not args.bf16 and not lean:
                try:
                    r10 = run_schnet4096(a16, rank, world, dev, mdist, False, steps=28, warmup=6)
                    sec["schnet4096"]["steps10"] = {"value": r10["value"], "ms_per_step": r10["ms_per_step"], "md_steps_per_pass": 10 * 8,
                                                    "searches_per_pass": (r10["config"].get("neighbour_list") or {}).get("searches_per_pass"),
                                                    "step_roof_frac": r10["roofline"]["step_roof"]["frac"]}
                except (Exception, SystemExit) as e:
                    sec["schnet4096"]["steps10"] = {"error": "%s: %s" % (type(e).__name__, e)}
                try:
                    a32 = copy.copy(args)
                    a32.frames_override = 53
                    f32 = run_schnet4096(a32, rank, world, dev, mdist, False, steps=5, warmup=2)
                    sec["schnet4096"]["f32"] = {k: f32[k] for k in ("value", "ms_per_step", "dtype")}
                    sec["schnet4096"]["f32"]["step_roof_frac"] = f32["roofline"]["step_roof"]["frac"]
                    sec["schnet4096"]["f32"]["kernel_frac_of_f32_mfma_peak"] = f32["roofline"]["frac"]
                except (Exception, SystemExit) as e:
                    sec["schnet4096"]["f32"] = {"error": "%s: %s" % (type(e).__name__, e)}
                # ... and bf16 MFMA operands with f32 node rows (rounds 2-5's schnet4096 leg)
                try:
                    a16p = copy.copy(a52)
                    a16p.bf16_rows = False
                    r16 = run_schnet4096(a16p, rank, world, dev, mdist, False, steps=8, warmup=3)
                    sec["schnet4096"]["bf16_f32rows"] = {k: r16[k] for k in ("value", "ms_per_step")}
                    sec["schnet4096"]["bf16_f32rows"]["step_roof_frac"] = r16["roofline"]["step_roof"]["frac"]
                    sec["schnet4096"]["bf16_f32rows"]["vs_f32"] = {k: v for k, v in (r16["config"].get("bf16_vs_f32") or {}).items() if k != "note"}
                except (Exception, SystemExit) as e:
                    sec["schnet4096"]["bf16_f32rows"] = {"error": "%s: %s" % (type(e).__name__, e)}
            if "water192" in sec and "error" not in sec["water192"]:
                try:                       # config #3's box the many-replica way
                    sec["water192x64"] = run_water192_stacked(args, rank, world, dev, mdist, R=64, steps=10, warmup=3)
                except (Exception, SystemExit) as e:
                    sec["water192x64"] = {"error": "%s: %s" % (type(e).__name__, e)}
            if "water192" in sec and "error" not in sec["water192"]:
                try:
                    wb = run_water192(args, rank, world, dev, mdist, False, steps=20, warmup=4, bf16=True)
                    sec["water192"]["bf16"] = {"value": wb["value"], "ms_per_step": wb["ms_per_step"],
                                               "parity_reference_golden": {k: v for k, v in (wb["config"].get("parity_reference_golden") or {}).items() if k != "vs"}}
                except (Exception, SystemExit) as e:
                    sec["water192"]["bf16"] = {"error": "%s: %s" % (type(e).__name__, e)}
            out["secondary"] = sec
            pending, _DEFERRED = _DEFERRED, None
            for fn in pending:
                try:
                    fn()
                except (Exception, SystemExit) as e:        # (a CPU leg must not take the measured lines down)
                    out.setdefault("cpu_leg_errors", []).append("%s: %s" % (type(e).__name__, e))
            # the other north-star workloads as flat scalars the driver's parser keeps, and one line each
            for name, rec in sec.items():
                out["config"].update(_flat(name, rec))
                lines.append(_line(name, rec))
            wbf = (sec.get("water192") or {}).get("bf16") or {}
            if "value" in wbf:
                out["config"]["water192_bf16_md_steps_per_s"] = float("%.6g" % wbf["value"])
                out["config"]["water192_bf16_parity_max_abs_dq"] = (wbf.get("parity_reference_golden") or {}).get("max_abs_dq")
            s4 = sec.get("schnet4096") or {}
            if "error" not in s4:
                v32 = (s4.get("config") or {}).get("bf16_vs_f32") or {}
                for k, kk in (("max_abs_dq_A", "schnet4096_52step_vs_f32_max_abs_dq"), ("dtheta_rel_to_largest", "schnet4096_52step_vs_f32_rel_dtheta")):
                    if k in v32:
                        out["config"][kk] = float("%.4g" % v32[k])
                for k, tag in (("f32", "schnet4096_f32"), ("bf16_f32rows", "schnet4096_bf16_f32rows"), ("steps10", "schnet4096_10step")):
                    if isinstance(s4.get(k), dict) and "value" in s4[k]:
                        out["config"][tag + "_md_steps_per_s"] = float("%.6g" % s4[k]["value"])
                        out["config"][tag + "_step_roof_frac"] = s4[k].get("step_roof_frac")
                ss = (s4.get("config") or {}).get("single_system") or {}
                if "md_steps_per_s" in ss:
                    out["config"]["single_system_md_steps_per_s"] = float("%.6g" % ss["md_steps_per_s"])
                    out["config"]["single_system_us_per_md_step"] = float("%.6g" % ss["us_per_md_step"])
                    for k in ("launches_per_md_step", "busy_us_per_md_step", "ms_per_pass", "kernel_frac"):
                        if k in ss:
                            out["config"]["single_system_" + k] = ss[k]
    if rank == 0:
        out["config"] = _ordered_config(out["config"])
        _write_detail(out)
        for ln in lines:
            print(ln, flush=True)
        if args.full_line:
            print(json.dumps(out), flush=True)
        else:
            head = {k: v for k, v in out.items() if k != "secondary"}
            print(json.dumps(_compact(head, cut=110)), flush=True)
    import torch.distributed as tdist
    if tdist.is_available() and tdist.is_initialized():
        mdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
