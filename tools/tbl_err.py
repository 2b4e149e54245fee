"""Accuracy of the tabulated pair path (MDG_PAIR_TABLE) against the reference golden as a function of the
node count: python tools/tbl_err.py   (MI355X)"""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conftest import load_golden
import test_gpu_parity as tp
from mdgrad_amd.observable import rdf
from mdgrad_amd.sovlers import odeint_adjoint
g = load_golden("pair_mlp")
for nodes, rmin in [(512, 0.2), (1024, 0.2), (2048, 0.2), (4096, 0.2), (1024, 0.3)]:
    system, mlp, prior, integ = tp._pair_mlp_setup(g, analytic=True)
    integ.table_nodes, integ.table_rmin = nodes, rmin
    y0 = [s_.clone().requires_grad_(True) for s_ in integ.get_inital_states(wrap=True)]
    t = torch.Tensor([float(g["dt"]) * i for i in range(9)]).to("cuda:0")
    v_t, q_t, pv_t = odeint_adjoint(integ, tuple(y0), t, method="NH_verlet")
    _, _, gr = rdf(system, nbins=60, r_range=(0.75, 2.4))(q_t)
    ((gr - 1).pow(2).mean() + 0.01 * v_t[-1].pow(2).sum()).backward()
    gm = torch.cat([(p_.grad if p_.grad is not None else torch.zeros_like(p_)).reshape(-1) for p_ in mlp.parameters()]).cpu().numpy()
    gp = torch.cat([p_.grad.reshape(-1) for p_ in prior.parameters()]).cpu().numpy()
    print(nodes, rmin, "q err %.2e" % np.abs(q_t.detach().cpu().numpy() - g["q_t"]).max(),
          "gmlp err %.3e (max %.3e)" % (np.abs(gm - g["grad_mlp"]).max(), np.abs(g["grad_mlp"]).max()),
          "gprior err", np.abs(gp - g["grad_prior"]), g["grad_prior"],
          "gq0 err %.2e" % np.abs(y0[1].grad.cpu().numpy() - g["grad_q0"]).max())
