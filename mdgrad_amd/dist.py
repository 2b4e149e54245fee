"""Replica sharding across the GPUs of one node.

The reference has no distributed code: its training loop walks a list of independent
simulations sequentially (demo/fit_rdf_gnn.py:386-399).  Here each rank (one process per GPU)
owns a contiguous shard of the replica list, runs forward + adjoint with no communication,
and the only exchange is ONE all-reduce(SUM) of the flat parameter-gradient buffer per outer
step (RCCL over xGMI; backend "nccl" on ROCm), after which every rank applies the identical
optimizer step.  The same code runs on CPU tensors with the gloo backend (tests).
"""
import os
import weakref

import torch
import torch.distributed as dist


def init(backend=None, device_index=None):
    """Initialise torch.distributed from the torchrun environment (RANK, WORLD_SIZE, MASTER_*).
    Returns (rank, world_size, device).  world_size 1 without a launcher needs no rendezvous."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    use_gpu = torch.cuda.is_available()
    if backend is None:
        # MDG_DIST_BACKEND=gloo lets several ranks share one GPU (control-flow tests on a 1-GPU box;
        # RCCL refuses two ranks on one device)
        backend = os.environ.get("MDG_DIST_BACKEND", "nccl" if use_gpu else "gloo")
    device = torch.device("cpu")
    if use_gpu:
        idx = local if device_index is None else device_index
        if os.environ.get("MDG_SINGLE_DEVICE") == "1":
            idx = 0
        torch.cuda.set_device(idx)
        device = torch.device("cuda", idx)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, device


def shard_range(n_items, rank, world):
    """Contiguous block partition of range(n_items): first (n % world) ranks get one extra."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def flatten_grads(params):
    """Flat fp32 buffer of the parameter gradients in parameter order (the flattening of
    torchmd/tinydiffeq.py:106-108); missing gradients contribute zeros."""
    params = list(params)
    if not params:
        return torch.zeros(0)
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])


def unflatten_to_grads(flat, params):
    pos = 0
    for p in params:
        n = p.numel()
        p.grad = flat[pos:pos + n].reshape(p.shape).clone()
        pos += n


class GradBucket:
    """ONE persistent flat gradient buffer for a parameter list; every `p.grad` is a view into it.

    The buffer is allocated once (7 MB for the SchNet of BASELINE config #5) and handed to the collective as is:
    no `torch.cat`, no per-parameter clone per outer step.  A gradient that autograd (re)allocated since the last
    step -- `zero_grad(set_to_none=True)` drops the views -- is copied into its slice once and `p.grad` re-pointed
    at the view, so the optimizer reads the reduced values; gradients accumulated in place into the views (the
    `zero_grad(set_to_none=False)` idiom) cost nothing."""

    def __init__(self, params):
        params = list(params)
        self._refs = [weakref.ref(p) for p in params]     # (weak: the bucket must not keep a model alive, nor form a cycle
        self.offsets = []                                 #  parameter -> bucket -> parameter -- ADVICE r4)
        n = 0
        for p in params:
            self.offsets.append(n)
            n += p.numel()
        ref = params[0]
        if any(p.dtype != ref.dtype or p.device != ref.device for p in params):
            raise ValueError("GradBucket needs parameters of one dtype on one device (got %s)"
                             % sorted({(str(p.dtype), str(p.device)) for p in params}))
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        self.views = [self.flat[o:o + p.numel()].view(p.shape) for o, p in zip(self.offsets, params)]

    @property
    def params(self):
        ps = [r() for r in self._refs]
        if any(p is None for p in ps):
            raise RuntimeError("GradBucket: a parameter of this bucket no longer exists")
        return ps

    def matches(self, params):
        return len(params) == len(self._refs) and all(r() is b for r, b in zip(self._refs, params))

    def gather(self):
        """Bring every parameter's gradient into the flat buffer (no-op for those that already live there)."""
        src, dst = [], []
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr() or g.shape != v.shape or not g.is_contiguous():
                src.append(g.detach().to(v.dtype))
                dst.append(v)
            if g is None or g is not v:
                p.grad = v
        if src:
            torch._foreach_copy_(dst, src)
        return self.flat


_buckets = {}          # id(first parameter) -> bucket, dropped by a finalizer when that parameter is collected


def _bucket_for(params):
    """The bucket of this parameter list, kept in a module-level table under the identity of the list's first parameter and
    removed when that parameter dies (weakref.finalize) -- nothing is attached to the Parameter itself, so torch.save /
    copy.deepcopy of a model do not drag the flat buffer along, and the bucket holds its parameters weakly."""
    params = list(params)
    key = id(params[0])
    b = _buckets.get(key)
    if b is not None and b.matches(params):
        return b
    b = GradBucket(params)
    if key not in _buckets:
        weakref.finalize(params[0], _buckets.pop, key, None)
    _buckets[key] = b
    return b


def _all_reduce(t, op):
    """all_reduce that also works when a CPU-only backend (gloo) is given a HIP tensor."""
    if t.is_cuda and dist.get_backend() == "gloo":
        c = t.cpu()
        dist.all_reduce(c, op=op)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op)


def all_reduce_grads(params, average=False):
    """One collective per outer step: SUM (or mean) of the flat gradient over all ranks; afterwards every
    `p.grad` is a view of the reduced persistent buffer (`GradBucket`).  No-op for world_size 1."""
    params = list(params)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1 or not params:
        return
    bucket = _bucket_for(params)
    flat = bucket.gather()
    _all_reduce(flat, dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    _all_reduce(t, dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    _all_reduce(t, dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(value, device):
    """[value of rank 0, ..., value of rank world-1] on every rank (one small all_gather)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    w = dist.get_world_size()
    gloo_on_gpu = device.type == "cuda" and dist.get_backend() == "gloo"
    t = torch.tensor([float(value)], dtype=torch.float64, device="cpu" if gloo_on_gpu else device)
    out = [torch.zeros_like(t) for _ in range(w)]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def collective_evidence(device, n_elems, reps=20):
    """What a record needs to show that the collective really ran over `world` ranks: the backend, the number of ranks
    that contributed to a SUM of ones, and the time of one all-reduce of an `n_elems` fp32 buffer (the size of the
    flat parameter gradient of the workload), measured on this rank around `reps` back-to-back collectives."""
    import time
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return {"backend": None, "world": 1, "ranks_seen": 1, "allreduce_us": None, "allreduce_elems": int(n_elems)}
    one = torch.ones(1, dtype=torch.float32, device=device)
    _all_reduce(one, dist.ReduceOp.SUM)
    buf = torch.zeros(max(1, int(n_elems)), dtype=torch.float32, device=device)
    _all_reduce(buf, dist.ReduceOp.SUM)
    if device.type == "cuda":
        torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        _all_reduce(buf, dist.ReduceOp.SUM)
    if device.type == "cuda":
        torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    return {"backend": dist.get_backend(), "world": dist.get_world_size(), "ranks_seen": int(round(float(one.item()))),
            "allreduce_us": max_over_ranks(us, device), "allreduce_elems": int(n_elems)}


def self_launch(script, argv, n_ranks):
    """Start `n_ranks` ranks of `script` through torch.distributed.run (one per visible GPU, rendezvous on 127.0.0.1
    at a free port) when the caller was started as a plain `python script --gpus N` without a launcher, and return the
    launcher's exit code.  The children see RANK / LOCAL_RANK / WORLD_SIZE and take the normal path."""
    import socket
    import subprocess
    import sys
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    rc = 1
    for _ in range(3):                    # a probed port can be taken between the probe and the rendezvous
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
               "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
        r = subprocess.run(cmd, env=env, stderr=subprocess.PIPE, text=True)
        sys.stderr.write(r.stderr)
        rc = r.returncode
        if rc == 0 or "address already in use" not in r.stderr.lower():
            break
    return rc
