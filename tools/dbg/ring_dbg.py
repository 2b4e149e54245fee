import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from conftest import load_golden
from test_gpu_parity import T, lj_setup, DEV
from mdgrad_amd import ops
g = load_golden("nhc_traj_lj")
system, mdl, integ = lj_setup(g)
R = 4
rng = np.random.default_rng(3)
pos = np.mod(g["pos"][None] + rng.normal(0, 0.03, (R,) + g["pos"].shape), g["cell"]).astype(np.float32)
vel = rng.normal(0, 1.0, pos.shape).astype(np.float32)
for nT in (2, 12):
    res = {}
    for block in (128, 64):
        spec = integ.fused_spec("NH_verlet"); spec.block = block
        t = torch.Tensor([0.005 * i for i in range(nT)])
        v0, q0 = T(vel, DEV).requires_grad_(True), T(pos, DEV).requires_grad_(True)
        pv0 = torch.zeros(R, 5, device=DEV, requires_grad=True)
        v_t, q_t, pv_t = ops.FusedTrajFn.apply(v0, q0, pv0, t.to(DEV), spec.flat_params(), spec)
        mdl.zero_grad()
        (q_t.pow(2).sum() / 100 + v_t[:, -1].pow(2).sum() + pv_t[:, -1].sum()).backward()
        res[block] = [x.detach().cpu() for x in (v_t, q_t, pv_t, v0.grad, q0.grad, pv0.grad, mdl.sigma.grad, mdl.epsilon.grad)]
    print("nT", nT)
    for name, a, b in zip("v_t q_t pv_t gv0 gq0 gpv0 gsig geps".split(), res[128], res[64]):
        print("  %-5s max|old| %.4g  max diff %.4g  nan(new) %d" % (name, float(a.abs().max()), float((a - b).abs().nan_to_num(1e9).max()), int(torch.isnan(b).sum())))
