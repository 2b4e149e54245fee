"""Cell-sweep RDF: fine z-bins (round 6, csrc/rdf_cell.hip make_grid: +- zw bins of a column instead of three bins of >= the
cutoff) against MDG_RDF_CELL_ZFINE=0 on the 4 096-atom liquid at the lj4096 bench shape (704 frames) -- histogram counts,
gradient deviation, HIP-event times."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from bench import lj_liquid
from mdgrad_amd.observable import rdf
from mdgrad_amd.system import System, Atoms

dev = "cuda:0"
rng = np.random.default_rng(5)
pos, L = lj_liquid(16, 0.845, rng)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 704
frames = np.stack([np.mod(pos + rng.normal(0, 0.06, pos.shape), L) for _ in range(8)]).astype(np.float32)
frames = np.concatenate([frames] * (F // 8))
system = System(Atoms(positions=pos, cell=[L, L, L], numbers=np.ones(len(pos))), device=dev)
obs = rdf(system, nbins=100, r_range=(0.75, 2.5))
wgt = torch.linspace(1, -1, 100, device=dev)
res = {}
for zf in ("1", "0"):
    os.environ["MDG_RDF_CELL_ZFINE"] = zf
    x = torch.from_numpy(frames).to(dev).requires_grad_(True)
    best = [1e9, 1e9]
    for rep in range(5):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        count, _, gr = obs(x)
        e[1].record()
        (gx,) = torch.autograd.grad((gr * wgt).sum(), x)
        e[2].record()
        torch.cuda.synchronize()
        best = [min(best[0], e[0].elapsed_time(e[1])), min(best[1], e[1].elapsed_time(e[2]))]
    res[zf] = (count.clone(), gx.clone())
    print("zfine=%s frames=%d fwd %.3f ms bwd %.3f ms" % (zf, F, best[0], best[1]))
print("histogram equal:", bool(torch.equal(res["1"][0], res["0"][0])), " max |d count|:", float((res["1"][0] - res["0"][0]).abs().max()))
d = (res["1"][1] - res["0"][1]).abs()
print("gradient max dev / max entry: %.3g" % (float(d.max()) / float(res["0"][1].abs().max())))
