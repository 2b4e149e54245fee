import cProfile, pstats, io, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mdgrad_amd import potentials as P, units
from mdgrad_amd.interface import PairPotentials, GNNPotentials, Stack
from mdgrad_amd.md import NoseHooverChain
from mdgrad_amd.nn import get_model
from mdgrad_amd.observable import rdf
from mdgrad_amd.system import System, Diamond
from mdgrad_amd.sovlers import odeint_adjoint
dev="cuda:0"; rng=np.random.default_rng(0)
a = units.get_unit_len(0.997, 18.01528, 8); size=4
atoms = Diamond("O",(size,)*3,a); atoms.set_positions(np.mod(atoms.get_positions()+rng.normal(0,0.2,(len(atoms),3)),a*size)); atoms.masses[:]=18.01528
system=System(atoms,device=dev); kT=298*units.kB; system.set_temperature(kT,rng=rng)
torch.manual_seed(0)
net=get_model({"n_atom_basis":64,"n_filters":128,"n_gaussians":30,"n_convolutions":2,"cutoff":6.0})
integ=NoseHooverChain(Stack({"gnn":GNNPotentials(system,net,cutoff=6.0),"prior":PairPotentials(system,P.ExcludedVolume(2.6,0.01,12),cutoff=6.0)}),system,T=kT,num_chains=5,Q=50.0).to(dev)
t=torch.Tensor([units.fs*i for i in range(6)]).to(dev)
def once(prof=None):
    y0=tuple(integ.get_inital_states(wrap=True))
    traj=odeint_adjoint(integ,y0,t,method="NH_verlet")
    loss=traj[1].pow(2).mean()
    torch.cuda.synchronize()
    if prof: prof.enable()
    t0=time.perf_counter(); loss.backward(); torch.cuda.synchronize(); t1=time.perf_counter()
    if prof: prof.disable()
    return t1-t0
for _ in range(3): once()
print("backward s:", once())
from mdgrad_amd import sovlers
orig = sovlers._analytic_nhc_adjoint
pr=cProfile.Profile()
def wrapped(*a, **k):
    pr.enable(); r = orig(*a, **k); torch.cuda.synchronize(); pr.disable(); return r
sovlers._analytic_nhc_adjoint = wrapped
print("profiled:", once())
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats("tottime").print_stats(45); print(s.getvalue()[:9000])
