"""Verlet reuse of the fixed-capacity SchNet / pair lists (ops.VerletList, mdg_nbr_verlet_rebuild): the list is searched
with a skin and kept while no atom has moved more than half of it; the consumers re-apply the list builders' exact cutoff
test per pair (mdg_edge_geom_masked -> d = -1 -> the cfconv kernels skip the slot; mdg_pair_eval_ell_into's recheck bit).
Every evaluation must see the pair set -- and, up to summation order, the numbers -- of a fresh search at the cutoff
(the reference rebuilds at every call: torchmd/md.py:200-204, topology.py:30-73)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from test_gpu_parity import T, close, mk_system, DEV
from test_gpu_fused_block import _setup

pytestmark = pytest.mark.gpu


def _rel(a, b, what, tol=2e-5):
    close(a, b, 0, tol * float(b.abs().max()) + 1e-7, what)


@pytest.mark.parametrize("n_side,bf16", [(6, False), (10, False), (10, True)])
def test_stored_list_with_skin_gives_the_fresh_lists_results(n_side, bf16):
    """216 atoms (all-pairs search) and 1 000 atoms (cell list): positions move inside the half-skin ball -> no new search,
    masked kernels == kernels on a fresh exact list (forward, tangent, reverse sweep with parameter gradients, geometry
    scatter, pair term); one atom leaves the ball -> the device decides to search again."""
    from mdgrad_amd import ops, _lib, potentials as P
    G, F, cutoff, skin = 30, 128, 5.0, 0.3
    x0, topo0, net = _setup(G, F, seed=5 + n_side, n_side=n_side, cutoff=cutoff)
    N = topo0.n_atoms
    L = 2.9 * n_side
    cs = _lib.make_cell(np.array([L, L, L], dtype=np.float32))
    longest = int(topo0.ell.cnt.max())
    vl = ops.VerletList(N, N, cs, cutoff, skin, None, (int(longest * 1.4) + 15) // 8 * 8, (int(topo0.n_edges * 1.4) + 1023) // 1024 * 1024, DEV)
    assert vl.use_cell == (n_side == 10)
    need = torch.zeros(2, dtype=torch.int32, device=DEV)
    vl.rebuild(x0, need)
    assert vl.builds() == 1 and need.tolist()[0] <= vl.max_nbr and need.tolist()[1] <= vl.capacity
    assert torch.equal(vl.pos_build, x0)
    rng = np.random.default_rng(n_side)
    step = rng.normal(0, 1, (N, 3))
    step = 0.12 * step / np.linalg.norm(step, axis=1)[:, None] * rng.uniform(0.2, 1.0, (N, 1))     # |move| < 0.15 = skin / 2
    x1 = x0 + T(step.astype(np.float32), DEV)
    vl.rebuild(x1, need)
    assert vl.builds() == 1, "every atom is inside its half-skin ball: the stored list must be reused"
    exact = ops.GraphTopo(ops.build_ell(x1, cs, cutoff))
    assert int(vl.n_valid) > exact.n_edges, "the stored list carries the skin's extra candidates"
    fn = ops.FilterNet(*net, bf16=bf16)
    w = torch.randn(N, 3, device=DEV)
    h, hd, mb, mdb = [torch.randn(N, F, device=DEV) for _ in range(4)]

    def sweep(topo):
        d, uhat, dd, ddel = ops.edge_geom(x1, topo, w)
        m, md, hs, hds = ops.cfconv_fwd(fn, d, dd, h, hd, topo, want_sums=True)
        both = torch.zeros(2, topo.n_edges, device=DEV)
        th = ops.cfconv_bwd(fn, d, dd, topo, h, hd, mb, mdb, both[0], both[1], want_theta=True)
        Fv, dwf = ops.edge_geom_bwd(both[0], both[1], d, dd, uhat, ddel, topo)
        plain = torch.zeros(topo.n_edges, device=DEV)
        ops.cfconv_bwd(fn, d, None, topo, h, None, None, mdb, None, plain)
        F1, _ = ops.edge_geom_bwd(None, plain, None, None, uhat, None, topo)
        return d, [m, md, hs, hds, th[0], th[1], th[2], Fv, dwf, F1]

    d_v, got = sweep(vl.topo)
    d_e, ref = sweep(exact)
    assert int((d_v[: int(vl.n_valid)] >= 0).sum()) == exact.n_edges, "the masked pair set is the fresh search's pair set"
    tol = 2e-5 if not bf16 else 2e-3         # (bf16: the operand rounding sees the same values; the tile grouping differs)
    for a, b, nm in zip(got, ref, ("m", "md", "hsum", "hdsum", "gW1", "gb1", "gW2", "F", "d(w.F)/dx", "F (plain sweep)")):
        _rel(a, b, "stored list vs fresh list: " + nm, tol)
    # pair term over the stored rows with the exact cutoff re-applied
    term = ops.make_term(P.LennardJones(1.0, 1.0).mdg_term(), cutoff, 0, 2, None)
    theta = torch.tensor([1.0, 1.0], device=DEV)
    o_v = ops.pair_eval(vl.ell, x1, term, theta, w=w, energy=True, grad=True)
    o_e = ops.pair_eval(exact.ell, x1, term, theta, w=w, energy=True, grad=True)
    for k in ("energy", "grad", "hw", "gtheta", "gtheta_w"):
        _rel(o_v[k], o_e[k], "pair term over the stored list: " + k)
    # one atom leaves its ball: the device asks for a new search, at the new positions
    x2 = x1.clone()
    x2[N // 2] = x0[N // 2] + torch.tensor([0.2, 0.0, 0.0], device=DEV)       # 0.2 from where the list was built > skin / 2
    vl.rebuild(x2, need)
    assert vl.builds() == 2 and torch.equal(vl.pos_build, x2)
    fresh = ops.GraphTopo(ops.build_ell(x2, cs, cutoff + skin))
    assert int(vl.n_valid) == fresh.n_edges and torch.equal(vl.nbr[: fresh.n_edges], fresh.nbr)
    assert torch.equal(vl.cnt, fresh.ell.cnt)


def test_gnn_trajectory_reuses_its_list_and_matches_the_exact_list_path():
    """Stack(SchNet + prior) on the 64-bead CG box, 12 steps + adjoint through HIP-graph replay: far fewer searches than force
    evaluations, the same trajectory and gradients as the eager pass on exact lists rebuilt at every call, and as with the
    reuse switched off."""
    from mdgrad_amd import graphs
    from test_gpu_schnet import _gnn_integrator, _traj_and_grads
    g = load_golden("gnn_traj")
    system = mk_system(g["pos"], g["cell"], g["vel"], g["masses"], g["numbers"])
    t = torch.Tensor([float(g["dt"]) * i for i in range(13)]).to(DEV)
    integ = _gnn_integrator(g, system)
    assert graphs.enabled(integ)
    gnn = integ.model.models["gnn"]
    assert gnn.verlet_skin > 0
    for m in integ.model.models.values():        # (this box moves ~0.05 A per evaluation under its random-init forces: a 1 A skin)
        m.verlet_skin = 0.2
    out = _traj_and_grads(integ, system, t)
    vl = gnn._static["verlet"]
    evaluations = 3 * 12 + 4
    assert 1 <= vl.builds() < evaluations // 2, "searches %d of ~%d evaluations" % (vl.builds(), evaluations)
    integ.use_graphs = False
    ref = _traj_and_grads(integ, system, t)                       # eager: exact-size lists, rebuilt at every evaluation
    for a, b, name in zip(out, ref, ("v_t", "q_t", "pv_t", "dL/dtheta")):
        close(a, b, 1e-4, 2e-5 * float(b.abs().max()) + 1e-7, "stored lists vs exact lists: " + name)
    integ2 = _gnn_integrator(g, system)
    for m in integ2.model.models.values():
        m.verlet_skin = 0.0
    off = _traj_and_grads(integ2, system, t)
    assert integ2.model.models["gnn"]._static.get("verlet") is None
    for a, b, name in zip(out, off, ("v_t", "q_t", "pv_t", "dL/dtheta")):
        close(a, b, 1e-4, 2e-5 * float(b.abs().max()) + 1e-7, "reuse on vs off: " + name)


@pytest.mark.parametrize("how", ["fused_block_off", "wide_basis"])
def test_unfused_chain_never_sees_a_skin_list(how):
    """the unfused SchNet chain (`fused_block = False`, or n_gaussians > 64 -- the reference's search space
    reaches 320) reads every listed pair at full weight, so a list searched with cutoff + skin must not reach it: graph
    replay / sync-free passes give it the exact fixed-capacity list, and the trajectory + adjoint equal the eager pass on
    exact lists."""
    from mdgrad_amd import graphs, potentials as P
    from mdgrad_amd.interface import GNNPotentials, PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain
    from mdgrad_amd.nn import get_model, analytic
    from test_gpu_schnet import _traj_and_grads, params_of
    g = load_golden("gnn_traj")
    system = mk_system(g["pos"], g["cell"], g["vel"], g["masses"], g["numbers"])
    torch.manual_seed(5)
    prm = dict(params_of(g))
    if how == "wide_basis":
        prm["n_gaussians"] = 80
    net = get_model(prm)
    if how == "fused_block_off":
        net.fused_block = False
    gnn = GNNPotentials(system, net, cutoff=float(g["cutoff"]))
    prior = PairPotentials(system, P.ExcludedVolume(float(g["prior_sigma"]), float(g["prior_epsilon"]), 12), cutoff=float(g["cutoff"]))
    integ = NoseHooverChain(Stack({"gnn": gnn, "prior": prior}), system, T=float(g["T"]), num_chains=int(g["chains"]),
                            Q=float(g["Q"]), adjoint=True).to(DEV)
    assert gnn.supports_force_vjp() and not analytic.fused_ok(net)
    for m in integ.model.models.values():        # a skin wide enough that pairs in (rc, rc + skin] certainly exist
        m.verlet_skin = 0.2
    t = torch.Tensor([float(g["dt"]) * i for i in range(6)]).to(DEV)
    out = _traj_and_grads(integ, system, t)
    assert gnn._static is None or gnn._static.get("verlet") is None, "a skin list reached the unfused chain"
    integ.use_graphs = False
    for m in integ.model.models.values():
        m.verlet_skin = 0.0
    ref = _traj_and_grads(integ, system, t)
    for a, b, name in zip(out, ref, ("v_t", "q_t", "pv_t", "dL/dtheta")):
        close(a, b, 1e-4, 2e-5 * float(b.abs().max()) + 1e-7, "unfused chain, static lists vs exact lists: " + name)
