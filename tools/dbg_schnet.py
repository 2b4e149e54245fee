import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mdgrad_amd import ops, potentials as P, units
from mdgrad_amd.interface import GNNPotentials, PairPotentials, Stack
from mdgrad_amd.md import NoseHooverChain
from mdgrad_amd.nn import get_model
from mdgrad_amd.observable import rdf
from mdgrad_amd.sovlers import odeint_adjoint
from mdgrad_amd.system import System, Diamond
dev="cuda:0"; rng=np.random.default_rng(2000)
a=units.get_unit_len(0.997,18.01528,8); size=8
atoms=Diamond("O",(size,)*3,a); atoms.masses[:]=18.01528
base=System(atoms,device=dev); system=base.replicate(4); L=a*size
system.set_positions(np.mod(system.get_positions()+rng.normal(0,0.2,(len(system),3)),L))
kT=298*units.kB; system.set_temperature(kT,rng=rng); torch.manual_seed(0)
net=get_model({"n_atom_basis":64,"n_filters":128,"n_gaussians":30,"n_convolutions":2,"cutoff":6.0})
gnn=GNNPotentials(system,net,cutoff=6.0)
integ=NoseHooverChain(Stack({"gnn":gnn,"prior":PairPotentials(system,P.ExcludedVolume(2.6,0.01,12),cutoff=6.0)}),system,T=kT,num_chains=5,Q=50.0).to(dev)
print("E init", gnn.inputs["_topo"].n_edges, "max_nbr", gnn.inputs["_topo"].ell.max_nbr)
t=torch.Tensor([units.fs*i for i in range(11)]).to(dev)
y0=tuple(integ.get_inital_states(wrap=True))
v_t,q_t,pv_t=odeint_adjoint(integ,y0,t,method="NH_verlet")
print("E after fwd", gnn.inputs["_topo"].n_edges, "finite", bool(torch.isfinite(q_t).all()), float(q_t.abs().max()))
obs=rdf(system,nbins=60,r_range=(2.0,6.0))
loss=(obs(q_t[::5])[2]-1).pow(2).mean(); loss.backward()
print("E after bwd", gnn.inputs["_topo"].n_edges, "loss", float(loss))
g=torch.cat([p.grad.reshape(-1) for p in integ.parameters() if p.grad is not None]); print("grad finite", bool(torch.isfinite(g).all()), float(g.abs().max()))
print("cnt max", int(gnn.inputs["_topo"].ell.cnt.max()))
print("---- force check")
q0=y0[1].clone()
integ.update_topology(q0)
F_an=integ.model.force(q0)
qa=q0.clone().requires_grad_(True)
U=integ.model(qa); (g_,)=torch.autograd.grad(U.sum(),qa)
print("analytic finite", bool(torch.isfinite(F_an).all()), "autograd finite", bool(torch.isfinite(g_).all()))
print("max|F_an|", float(F_an[torch.isfinite(F_an)].abs().max()), "max|F_ag|", float(g_.abs().max()), "diff", float((F_an+g_)[torch.isfinite(F_an)].abs().max()))
bad=(~torch.isfinite(F_an)).any(1).nonzero().reshape(-1)
print("bad atoms", bad[:10].tolist(), len(bad))
Fg=gnn.force(q0); Fp=integ.model.models["prior"].force(q0)
print("gnn finite", bool(torch.isfinite(Fg).all()), "prior finite", bool(torch.isfinite(Fp).all()))
topo=gnn.inputs["_topo"]
print("nbr max", int(topo.nbr.max()), "eid max", int(topo.eid.max()), "E", topo.n_edges, "cnt sum/2", int(topo.ell.cnt.sum())//2)
