"""Multi-process (world_size 2, gloo, CPU) coverage of the replica-sharding layer that bench.py
and the training loop use on N GPUs with RCCL: shard partition, flat-gradient all-reduce, the
max/sum helpers.  The HIP kernels are not involved (no GPU here); the collective pattern is."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from mdgrad_amd import dist as mdist
    r, w, dev = mdist.init(backend="gloo")
    assert (r, w) == (rank, world) and dev.type == "cpu"
    # replica shard: 7 replicas over 2 ranks -> [0,4) and [4,7)
    lo, hi = mdist.shard_range(7, rank, world)
    # each replica r contributes gradient (r+1) * [1, 10] to a 2-parameter model (sigma, epsilon)
    sigma = torch.nn.Parameter(torch.tensor([1.0]))
    eps = torch.nn.Parameter(torch.tensor([1.0]))
    sigma.grad = torch.tensor([float(sum(k + 1 for k in range(lo, hi)))])
    eps.grad = torch.tensor([10.0 * sum(k + 1 for k in range(lo, hi))])
    mdist.all_reduce_grads([sigma, eps])
    mdist.barrier()
    mx = mdist.max_over_ranks(1.0 + rank, dev)
    sm = mdist.sum_over_ranks(1.0 + rank, dev)
    # identical optimizer step on every rank
    opt = torch.optim.SGD([sigma, eps], lr=0.01)
    opt.step()
    q.put((rank, lo, hi, float(sigma.grad), float(eps.grad), mx, sm, float(sigma), float(eps)))
    dist.destroy_process_group()


def test_replica_sharding_and_grad_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 4), (4, 7)]
    for r in res:
        assert r[3] == 28.0 and r[4] == 280.0            # sum over all 7 replicas on every rank
        assert r[5] == 2.0 and r[6] == 3.0
    assert res[0][7:] == res[1][7:]                       # parameters stay identical across ranks


def test_shard_range_partitions_exactly():
    from mdgrad_amd.dist import shard_range
    for n in (0, 1, 7, 8, 4096, 4099):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts[:-1], parts[1:]))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def test_flatten_roundtrip_and_world1_noop():
    from mdgrad_amd import dist as mdist
    ps = [torch.nn.Parameter(torch.randn(3, 2)), torch.nn.Parameter(torch.randn(5))]
    ps[0].grad = torch.randn(3, 2)
    flat = mdist.flatten_grads(ps)
    assert flat.shape == (11,) and torch.equal(flat[6:], torch.zeros(5))
    mdist.unflatten_to_grads(flat * 2, ps)
    assert torch.equal(ps[0].grad.reshape(-1), flat[:6] * 2)
    mdist.all_reduce_grads(ps)                             # not initialised: no-op
    assert mdist.max_over_ranks(3.5, torch.device("cpu")) == 3.5
