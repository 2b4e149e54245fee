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


def test_grad_bucket_keeps_one_flat_buffer_with_parameter_views():
    """all_reduce_grads' persistent buffer (mdgrad_amd.dist.GradBucket): gradients that autograd re-allocated are
    copied into their slice once and p.grad re-pointed at the view; in-place accumulation into the views is free;
    missing gradients are zeros."""
    from mdgrad_amd.dist import GradBucket
    ps = [torch.nn.Parameter(torch.randn(3, 2)), torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(1))]
    b = GradBucket(ps)
    ps[0].grad = torch.arange(6.0).reshape(3, 2)
    ps[2].grad = torch.tensor([7.0])
    flat = b.gather()
    assert flat.data_ptr() == b.flat.data_ptr()
    assert torch.equal(flat, torch.tensor([0, 1, 2, 3, 4, 5, 0, 0, 0, 0, 0, 7.0]))
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(ps, b.views))
    # autograd accumulates in place into the views (zero_grad(set_to_none=False) idiom): nothing to copy
    (ps[0].sum() * 2 + ps[1].sum()).backward()
    assert ps[0].grad.data_ptr() == b.views[0].data_ptr()
    flat2 = b.gather()
    assert torch.equal(flat2[:6], torch.arange(6.0) + 2) and torch.equal(flat2[6:11], torch.ones(5))
    # set_to_none=True drops the views: the next gather restores them and zeroes the missing ones
    for p in ps:
        p.grad = None
    ps[1].grad = torch.full((5,), 3.0)
    flat3 = b.gather()
    assert torch.equal(flat3, torch.tensor([0, 0, 0, 0, 0, 0, 3, 3, 3, 3, 3, 0.0]))
    flat3 *= 2                                        # what the collective does: the parameters see it through the views
    assert torch.equal(ps[1].grad, torch.full((5,), 6.0))


_SELF_LAUNCH_SCRIPT = '''
import argparse, os, sys
sys.path.insert(0, %r)
import torch
from mdgrad_amd import dist as mdist
ap = argparse.ArgumentParser(); ap.add_argument("--gpus", type=int, default=1); args = ap.parse_args()
if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    sys.exit(mdist.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))
rank, world, dev = mdist.init(backend="gloo")
p = torch.nn.Parameter(torch.zeros(3)); p.grad = torch.full((3,), float(rank + 1))
mdist.all_reduce_grads([p])
ev = mdist.collective_evidence(dev, 3, reps=2)
ms = mdist.gather_over_ranks(10.0 + rank, dev)
if rank == 0:
    print("RESULT", world, float(p.grad[0]), ev["backend"], ev["world"], ev["ranks_seen"], ev["allreduce_us"] > 0, ms, flush=True)
import torch.distributed as d
d.destroy_process_group()
'''


def test_plain_command_spawns_its_own_ranks(tmp_path):
    """`python script --gpus 2` without a launcher re-executes itself through torch.distributed.run (bench.py and
    examples/fit_rdf_gnn.py use the same helper): rank 0's line arrives on the parent's stdout, the collective sees 2
    ranks."""
    import subprocess
    script = tmp_path / "selflaunch.py"
    script.write_text(_SELF_LAUNCH_SCRIPT % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(script), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
    assert lines == ["RESULT 2 3.0 gloo 2 2 True [10.0, 11.0]"], r.stdout[-1000:]


def test_grad_bucket_does_not_hang_on_the_parameter_nor_keep_the_model_alive():
    """the flat gradient buffer is not an attribute of a Parameter (deepcopy / pickling of a model stay small), holds
    its parameters weakly, and its table entry goes when the model does."""
    import copy
    import gc
    import weakref
    import torch
    from mdgrad_amd import dist as mdist
    model = torch.nn.Linear(4, 3)
    params = list(model.parameters())
    b = mdist._bucket_for(params)
    assert mdist._bucket_for(params) is b
    assert not any(isinstance(v, mdist.GradBucket) for v in vars(params[0]).values())
    clone = copy.deepcopy(model)
    assert not any(isinstance(v, mdist.GradBucket) for p in clone.parameters() for v in vars(p).values())
    key, ref = id(params[0]), weakref.ref(params[0])
    del model, params, clone
    gc.collect()
    assert ref() is None and key not in mdist._buckets, "the bucket must not keep the parameters alive"
