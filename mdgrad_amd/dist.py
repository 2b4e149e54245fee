"""Replica sharding across the GPUs of one node.

The reference has no distributed code: its training loop walks a list of independent
simulations sequentially (demo/fit_rdf_gnn.py:386-399).  Here each rank (one process per GPU)
owns a contiguous shard of the replica list, runs forward + adjoint with no communication,
and the only exchange is ONE all-reduce(SUM) of the flat parameter-gradient buffer per outer
step (RCCL over xGMI; backend "nccl" on ROCm), after which every rank applies the identical
optimizer step.  The same code runs on CPU tensors with the gloo backend (tests).
"""
import os

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


def _all_reduce(t, op):
    """all_reduce that also works when a CPU-only backend (gloo) is given a HIP tensor."""
    if t.is_cuda and dist.get_backend() == "gloo":
        c = t.cpu()
        dist.all_reduce(c, op=op)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op)


def all_reduce_grads(params, average=False):
    """One collective per outer step: SUM (or mean) of the flat gradient over all ranks,
    written back into p.grad.  No-op for world_size 1."""
    params = list(params)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1 or not params:
        return
    flat = flatten_grads(params)
    _all_reduce(flat, dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    unflatten_to_grads(flat, params)


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
