"""Per-pass wall time and allocator statistics of the lj4096 bench workload (diagnostics)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from mdgrad_amd import ops, potentials as P
from mdgrad_amd.interface import PairPotentials, Stack
from mdgrad_amd.md import NoseHooverChain
from mdgrad_amd.observable import rdf
from mdgrad_amd.system import System, Atoms

R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
MODE = sys.argv[3] if len(sys.argv) > 3 else "direct"       # direct | list | noobs | gc
ops.RDF_CELL_DIRECT = MODE != "list"
import gc
dev = torch.device("cuda:0")
rng = np.random.default_rng(3000)
pos1, L = bench.lj_liquid(16, 0.845, rng)
N = len(pos1)
system = System(Atoms(positions=pos1, cell=[L, L, L], numbers=np.ones(N)), device=dev)
mdl = P.LennardJones(1.0, 1.0)
integ = NoseHooverChain(Stack({"pair": PairPotentials(system, mdl, cutoff=2.5)}), system, T=1.0, num_chains=5, Q=50.0).to(dev)
spec = integ.fused_spec("NH_verlet")
pos = torch.from_numpy(np.stack([bench.lj_liquid(16, 0.845, rng)[0] for _ in range(R)]).astype(np.float32)).to(dev)
vel = torch.from_numpy(rng.normal(0, 1.0, (R, N, 3)).astype(np.float32)).to(dev)
pv0 = torch.zeros(R, 5, device=dev)
obs = rdf(system, nbins=100, r_range=(0.75, 2.5))
target = torch.ones(100, device=dev)
t = torch.Tensor([0.005 * i for i in range(51)]).to(dev)
params = list(integ.parameters())
opt = torch.optim.Adam(params, lr=1e-4)
for k in range(K):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    v_t, q_t, pv_t = ops.fused_traj(vel, pos, pv0, t, spec.flat_params(), spec)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    loss = q_t[:, ::5].pow(2).mean() if MODE == "noobs" else (obs(q_t[:, ::5])[2] - target).pow(2).mean()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    opt.step()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    if MODE == "gc":
        del v_t, q_t, pv_t, loss
        a0 = torch.cuda.memory_allocated(); n = gc.collect(); a1 = torch.cuda.memory_allocated()
        print("   gc.collect: %d objects, allocated %.2f -> %.2f GB" % (n, a0 / 2**30, a1 / 2**30))
    st = torch.cuda.memory_stats()
    print("pass %2d  fwd %6.1f  rdf %6.1f  bwd %6.1f  opt %5.1f ms | reserved %.1f GB  dev_allocs %d  retries %d" % (
        k, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), st["reserved_bytes.all.current"] / 2**30,
        st["segment.all.allocated"], st["num_alloc_retries"]), flush=True)
print(ops.LARGE_STATS)
