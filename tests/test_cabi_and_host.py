"""CPU-only checks: the C-ABI library loads and exports every symbol include/mdgrad_hip.h
declares, argument validation fails loudly, and the host-side logic (System, masks, sharding,
wrap) behaves like the reference / the oracle."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "mdgrad_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mdg_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from mdgrad_amd import _lib
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), "libmdgrad_hip.so does not export %s" % s
    assert set(syms) == set(_lib.EXPORTED_SYMBOLS), "ctypes signatures out of sync with the header"
    assert lib.mdg_version() >= 100


def test_struct_layouts_match_header_sizes():
    from mdgrad_amd import _lib
    assert ctypes.sizeof(_lib.MdgPairTerm) == 48
    assert ctypes.sizeof(_lib.MdgTerms) == 8 + 4 * 48
    assert ctypes.sizeof(_lib.MdgCell) == 76
    assert ctypes.sizeof(_lib.MdgTrajParams) == 32 + 64


def test_argument_validation_is_loud():
    from mdgrad_amd import _lib
    lib = _lib.load()
    prm, cell, terms = _lib.MdgTrajParams(), _lib.make_cell([4.8] * 3), _lib.MdgTerms()
    prm.n_rep, prm.n_atoms, prm.n_frames, prm.n_chains, prm.ensemble = 1, 108, 10, 1, 0
    terms.n_terms = 1
    rc = lib.mdg_traj_fwd_small(ctypes.byref(prm), ctypes.byref(cell), ctypes.byref(terms),
                                None, None, None, None, None, None, None, None, None, None, None)
    assert rc == -1 and b"num_chains" in lib.mdg_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(rc, "mdg_traj_fwd_small")
    assert lib.mdg_nbr_build_dense(None, 0, ctypes.byref(cell), 1.0, None, None, None, None, 8, None, None) == -1


def test_no_cpu_fallback():
    from mdgrad_amd.system import System, FaceCenteredCubic
    from mdgrad_amd.interface import PairPotentials
    from mdgrad_amd.potentials import LennardJones
    system = System(FaceCenteredCubic("H", (2, 2, 2), 1.6), device="cpu")
    with pytest.raises(RuntimeError, match="HIP device"):
        PairPotentials(system, LennardJones(), cutoff=1.5)


def test_system_and_lattices():
    from mdgrad_amd.system import System, FaceCenteredCubic, Diamond
    from mdgrad_amd import units
    a = FaceCenteredCubic("H", (3, 3, 3), 1.6)
    ref, cell = O.fcc_lattice(3, 1.6)
    assert len(a) == 108 and np.allclose(a.get_positions(), ref) and np.allclose(np.diag(a.get_cell()), cell)
    d = Diamond("O", (2, 2, 2), 6.2)
    assert len(d) == 64 and np.allclose(d.get_positions(), O.diamond_lattice(2, 6.2)[0])
    assert np.allclose(d.get_masses(), 15.999)
    s = System(a, device="cpu")
    assert s.get_nxyz().shape == (108, 4) and s.get_batch()["num_atoms"].item() == 108
    assert np.allclose(s.get_cell_len(), 4.8) and abs(s.get_volume() - 4.8 ** 3) < 1e-9
    s.set_temperature(1.0, rng=np.random.default_rng(0))
    ke = 0.5 * (s.get_momenta() ** 2 / s.get_masses()[:, None]).sum()
    assert 0.5 * 324 * 0.7 < ke < 0.5 * 324 * 1.3
    v = np.random.default_rng(1).normal(size=(108, 3))
    s.set_velocities(v)
    assert np.allclose(s.get_velocities(), v)
    assert abs(units.get_unit_len(0.997, 18.01528, 8) - 6.2148) < 1e-3


def test_wrap_positions_matches_oracle():
    from mdgrad_amd.system import wrap_positions
    rng = np.random.default_rng(2)
    cell = np.array([[6.0, 0, 0], [1.2, 5.5, 0], [0.7, -0.9, 6.3]])
    pos = rng.normal(0, 8, (50, 3))
    assert np.allclose(wrap_positions(pos, cell), O.wrap_positions_oracle(pos, cell))
    w = wrap_positions(pos, np.array([4.0, 5.0, 6.0]))
    assert (w > -1e-6).all() and (w < np.array([4.0, 5.0, 6.0]) + 1e-6).all()


def test_masks_match_oracle():
    from mdgrad_amd.ops import build_mask
    from oracle.md_oracle import _pair_select_mask
    A, B = list(range(0, 20, 2)), list(range(1, 20, 3))
    ex = [[0, 1], [4, 7], [2, 10]]
    for it, e in [((A, B), None), (None, ex), ((A, A), ex), ((A, B), ex)]:
        m = build_mask(20, it, e, "cpu")
        assert torch.equal(m.bool(), _pair_select_mask(20, it, e))
    assert build_mask(20, None, None, "cpu") is None


def test_make_cell():
    from mdgrad_amd import _lib
    c = _lib.make_cell([4.8, 4.8, 4.8])
    assert c.diag == 1 and abs(c.inv[0] - np.float32(1 / np.float32(4.8))) < 1e-7
    t = _lib.make_cell(torch.tensor([[6.0, 0, 0], [1.2, 5.5, 0], [0.7, -0.9, 6.3]]))
    assert t.diag == 0
    h = np.array(list(t.h)).reshape(3, 3)
    assert np.allclose(h @ np.array(list(t.inv)).reshape(3, 3), np.eye(3), atol=1e-6)


def test_potential_descriptors():
    from mdgrad_amd import potentials as P, _lib
    assert P.LennardJones().mdg_term() == dict(kind=_lib.PAIR_LJ, p=12, q=6, c=1.0)
    assert P.LennardJones69().mdg_term()["p"] == 9
    assert P.ExcludedVolume(power=10).mdg_term() == dict(kind=_lib.PAIR_LJ, p=10, q=0, c=0.0)
    assert [n for n, _ in P.LJFamily().named_parameters()] == ["sigma", "epsilon"]
    r = torch.linspace(0.9, 2.0, 5)[:, None]
    for mod, kind, consts in [(P.LennardJones(1.1, 0.8), "lj", dict(p=12, q=6, c=1)),
                              (P.ExcludedVolume(1.0, 1.0, 12), "lj", dict(p=12, q=0, c=0)),
                              (P.ModifiedMorse(2.5, -1.2), "morse", dict(a=2.5, phi=-1.2)),
                              (P.Buck(100.0, 3.0, 2.0), "buck", {}), (P.Yukawa(1.3, 0.8), "yukawa", {})]:
        th = torch.cat([p.detach().reshape(-1) for p in mod.mdg_params()]) if mod.mdg_params() else torch.zeros(0)
        assert torch.allclose(mod(r).reshape(-1), O.pair_phi(kind, r.reshape(-1), th, consts)[0], rtol=1e-5)
    with pytest.raises(ValueError):
        P.LJFamily(rep_pow=12.5).mdg_term()


def test_torch_op_library_loads_and_registers_every_op():
    """libmdgrad_torch.so (csrc_torch/mdgrad_torch.cpp): TORCH_LIBRARY(mdgrad, ...) registers the op list of SURVEY 8b;
    there is no CPU implementation behind it (product path = HIP only)."""
    import torch
    from mdgrad_amd import _torch_ops
    ns = _torch_ops.get()
    assert ns is not None, "build it: python -c 'import __graft_entry__ as g; g.build()'"
    for op in _torch_ops.OPS:
        schema = str(getattr(ns, op).default._schema)
        assert schema.startswith("mdgrad::" + op + "("), schema
    with pytest.raises((RuntimeError, NotImplementedError)):
        ns.atb(torch.zeros(8, 4), torch.zeros(8, 4))


def test_struct_layout_of_the_fused_observable_descriptor():
    from mdgrad_amd import _lib
    assert ctypes.sizeof(_lib.MdgRdfFuse) == 8 + 7 * 4 + 4          # pointer + 7 words, padded to 8


def test_rdf_recognises_time_slices_of_a_fused_trajectory():
    """Host logic of the fused observable (observable.rdf._fused_raw): which views of a tagged trajectory tensor are
    'the frames start, start + stride, ... up to the last one' -- q_t, q_t[::k], q_t[s:], q_t[s::k], along the
    time axis the tag names -- and which are not (a cut-off end, a replica subset, a copy).  No kernel runs."""
    from mdgrad_amd import ops
    from mdgrad_amd.observable import rdf
    from mdgrad_amd.system import System, FaceCenteredCubic

    class Integ:                                  # (opted in: registration is off by default since round 5)
        fuse_observables = True
        _rdf_hint = None

    class Spec:                                   # what the observable touches on a FusedSpec
        rdf_hint = None

        def __init__(self):
            self._integrator = Integ()

    system = System(FaceCenteredCubic("H", (3, 3, 3), 1.6), device="cpu")
    obs = rdf(system, nbins=100, r_range=(0.75, 2.5))

    def hint_after(view_of, time_dim, shape):
        spec = Spec()
        q_t = torch.zeros(shape)
        ops.tag_trajectory(q_t, spec, None, time_dim)
        assert obs._fused_raw(view_of(q_t)) is None         # nothing cached yet: registers (or not) for the next launch
        return None if spec.rdf_hint is None else (spec.rdf_hint.start, spec.rdf_hint.stride)

    batched, single = (4, 12, 108, 3), (12, 108, 3)
    Integ.fuse_observables = False
    assert hint_after(lambda q: q, 1, batched) is None, "no opt-in: nothing is registered, the rdf stays a function of q_t"
    Integ.fuse_observables = True
    assert hint_after(lambda q: q, 1, batched) == (0, 1)
    assert hint_after(lambda q: q[:, ::3], 1, batched) == (0, 3)
    assert hint_after(lambda q: q[:, 2:], 1, batched) == (2, 1)
    assert hint_after(lambda q: q[:, 1::2], 1, batched) == (1, 2)
    assert hint_after(lambda q: q[::4], 0, single) == (0, 4)
    assert hint_after(lambda q: q[11:], 0, single) == (11, 1)
    assert hint_after(lambda q: q[:, :5], 1, batched) is None        # does not run to the last frame
    assert hint_after(lambda q: q[:, 1:9:2], 1, batched) is None
    assert hint_after(lambda q: q[:2], 1, batched) is None           # a replica subset
    assert hint_after(lambda q: q[:, :, :50], 1, batched) is None    # an atom subset
    assert hint_after(lambda q: q.clone(), 1, batched) is None       # a copy carries no tag
    # explicit registration on an integrator (what the observable otherwise does itself on first use)
    from mdgrad_amd import md
    integ = md._EOM()                              # (the integrators' base class: the method needs no device)
    integ.attach_observable(obs, start=5, stride=5)
    assert (integ._rdf_hint.start, integ._rdf_hint.stride) == (5, 5) and integ._rdf_hint.obs() is obs
    integ.attach_observable(None)
    assert integ._rdf_hint is None
    # a cached histogram is returned only to the observable and frame selection it was made for
    spec, q_t = Spec(), torch.zeros(batched)
    hint = ops.RdfFuse(obs, 0, 2)
    spec.rdf_hint = hint
    raw = torch.arange(100.0)
    ops.tag_trajectory(q_t, spec, raw, 1)
    assert obs._fused_raw(q_t[:, ::2]) is raw
    assert obs._fused_raw(q_t[:, ::3]) is None and spec.rdf_hint.stride == 3        # re-registered for the new selection
    other = rdf(system, nbins=50, r_range=(0.75, 2.5))
    ops.tag_trajectory(q_t, spec, raw, 1)
    spec.rdf_hint = hint
    q_t._mdg_traj = (spec, 1, hint, raw)
    assert other._fused_raw(q_t[:, ::2]) is None


def test_host_side_of_the_cell_sweep_rdf_and_the_large_path_workspace():
    """Host-only entry points (no kernel runs): which boxes the list-free RDF sweeps accept, their scratch size, loud
    argument checks, and the large-path workspace that holds the stored candidate lists (16-bit indices, 128 per atom
    and frame) until they would exceed the cap."""
    from mdgrad_amd import _lib
    lib = _lib.load()
    box = _lib.make_cell([16.9, 16.9, 16.9])
    assert lib.mdg_rdf_cell_supported(4096, ctypes.byref(box), 2.62) == 1
    assert lib.mdg_rdf_cell_supported(4096, ctypes.byref(box), 6.0) == 0                 # fewer than 3 bins per side
    assert lib.mdg_rdf_cell_supported(40000, ctypes.byref(box), 2.62) == 0               # above 32 768 atoms
    tri = _lib.make_cell(torch.tensor([[16.9, 0, 0], [3.0, 16.9, 0], [0, 0, 16.9]]))
    assert lib.mdg_rdf_cell_supported(4096, ctypes.byref(tri), 2.62) == 0                # orthorhombic cells only
    nb = int(16.9 // 2.62)
    # (round 6: fine z-bins -- as many as fit, at most 32 and 4 096 cells in all; MDG_RDF_CELL_ZFINE=0: bins of >= the cutoff)
    nbz = min(32, 4096 // (nb * nb))
    assert lib.mdg_rdf_cell_scratch(11, 4096, ctypes.byref(box), 2.62) == 5 * 11 * 4096 + 11 * (nb * nb * nbz + 1)
    os.environ["MDG_RDF_CELL_ZFINE"] = "0"
    try:
        assert lib.mdg_rdf_cell_scratch(11, 4096, ctypes.byref(box), 2.62) == 5 * 11 * 4096 + 11 * (nb ** 3 + 1)
    finally:
        del os.environ["MDG_RDF_CELL_ZFINE"]
    assert lib.mdg_rdf_cell_scratch(11, 4096, ctypes.byref(tri), 2.62) == -1
    assert lib.mdg_rdf_fwd_cell(None, 11, 4096, ctypes.byref(box), 2.62, None, 0.0175, -1632.0, 100, None, None, None) == -1
    assert b"rdf_fwd_cell" in lib.mdg_last_error()
    # workspace: running state + per-frame candidate rows (N * 128 * 2 bytes = N * 64 words) while they fit
    w10, w20 = lib.mdg_traj_large_workspace(1, 4096, 10, 2), lib.mdg_traj_large_workspace(1, 4096, 20, 2)
    per_frame = (w20 - w10) / 10
    # (+ per frame: the row counts, the build's permutation and its bin columns' first slots -- the column tiles)
    assert 4096 * 64 <= per_frame <= 4096 * 66 + 2048, per_frame
    huge = lib.mdg_traj_large_workspace(64, 16384, 2000, 2)                              # lists would need > 32 GiB
    assert huge < 64 * 16384 * 2000, "beyond the cap the lists are not kept (every evaluation searches)"


def test_host_side_of_the_stale_list_entry_points_of_the_large_path():
    """mdg_traj_*_large_stale without a GPU: the size of the persistent row buffer and the refusals that happen before any
    launch (frequency, counter, missing rows, a tabulated pair model)."""
    from mdgrad_amd import _lib, ops
    lib = _lib.load()
    assert lib.mdg_traj_large_stale_words(2, 100) == 2 * 100 * 257 and lib.mdg_traj_large_stale_words(0, 5) == -1
    prm, cell = _lib.MdgTrajParams(), _lib.make_cell([16.9] * 3)
    prm.n_rep, prm.n_atoms, prm.n_frames, prm.n_chains, prm.ensemble = 1, 4096, 5, 3, 0
    lj = ops.make_terms([ops.make_term(dict(kind=0,      # MDG_PAIR_LJ
                                             p=12, q=6, c=1.0), 2.5, 0, 2, None)], 2)
    fake = ctypes.c_void_p(0x1000)                       # (never dereferenced: every call below is refused during validation)
    B = ctypes.byref

    def fwd(terms, freq, count0, rows):
        return lib.mdg_traj_fwd_large_stale(B(prm), B(cell), B(terms), fake, fake, fake, fake, fake, fake, fake, fake, fake, fake,
                                            fake, freq, count0, rows, None)

    def adj(terms, freq, count0, rows):
        return lib.mdg_traj_adj_large_stale(B(prm), B(cell), B(terms), fake, fake, fake, fake, fake, fake, fake, fake, fake, fake,
                                            fake, fake, fake, fake, fake, freq, count0, rows, None)

    for call in (fwd, adj):
        assert call(lj, 0, 0, fake) == -1 and b"frequency" in lib.mdg_last_error()
        assert call(lj, 3, -1, fake) == -1 and b"counter" in lib.mdg_last_error()
        assert call(lj, 3, 0, None) == -1 and b"list buffer" in lib.mdg_last_error()
    tab = ops.make_terms([ops.make_term(dict(kind=ops.MDG_PAIR_TABLE, p=64, a=0.25, phi=0.1, c=1.0), 2.5, 0, 128, None)], 128)
    assert fwd(tab, 3, 0, fake) == -1 and b"tabulated" in lib.mdg_last_error()
    prm.n_atoms = 40000
    assert fwd(lj, 3, 0, fake) == -1                       # (beyond the large path's 32 768 atoms)


def test_host_side_of_the_row_chain_and_the_nh_half_step_scratch():
    """mdg_row_chain / mdg_nhv_scratch_floats without a GPU: struct layout, argument validation (every refusal happens before
    a launch), the empty call, and the size of the cross-workgroup scratch (partials of 1 024-element chunks, a ticket per
    replica, one ticket for the whole grid)."""
    from mdgrad_amd import _lib
    lib = _lib.load()
    assert ctypes.sizeof(_lib.MdgChainStage) == 15 * 8 + 6 * 4      # (13 f32 pointers, 2 bf16 mirrors, 6 ints)
    dummy = ctypes.c_void_p(0x1000)                      # never dereferenced: every call below returns before its launch
    st = (_lib.MdgChainStage * 2)()
    st[0].W, st[0].in0, st[0].K, st[0].M = dummy, dummy, 64, 32
    st[1].W, st[1].K, st[1].M = dummy, 32, 16
    assert lib.mdg_row_chain(st, 2, 0, 1, None) == 0                       # no rows: nothing to do
    assert lib.mdg_row_chain(st, 0, 16, 0, None) != 0 and b"stages" in lib.mdg_last_error()
    assert lib.mdg_row_chain(st, _lib.CHAIN_MAX_STAGES + 1, 16, 0, None) != 0
    st[1].K = 48                                                           # does not continue the previous stage's 32 outputs
    assert lib.mdg_row_chain(st, 2, 16, 0, None) != 0 and b"previous stage" in lib.mdg_last_error()
    st[1].K, st[1].mode = 32, _lib.CHAIN_HEAD                              # HEAD needs an activation and the readout row
    assert lib.mdg_row_chain(st, 2, 16, 0, None) != 0 and b"HEAD" in lib.mdg_last_error()
    st[1].mode, st[1].M = _lib.CHAIN_NONE, _lib.CHAIN_MAX_WIDTH + 1
    assert lib.mdg_row_chain(st, 2, 16, 0, None) != 0 and b"width" in lib.mdg_last_error()
    st[1].M = 16
    st[0].in0 = None                                                       # the first stage reads global memory
    assert lib.mdg_row_chain(st, 2, 16, 0, None) != 0
    # scratch of the mdg_nhv_* launches: 2 floats per (replica, chunk) + a ticket per replica + the grid's ticket
    assert lib.mdg_nhv_scratch_floats(1, 64) == 1 * 1 * 2 + 1 + 1
    assert lib.mdg_nhv_scratch_floats(8, 512) == 8 * 2 * 2 + 8 + 1         # 1 536 elements: two chunks
    assert lib.mdg_nhv_scratch_floats(1, 4096) == 12 * 2 + 1 + 1
    assert lib.mdg_nhv_scratch_floats(0, 64) == 0


def test_layer_parameter_cache_follows_replaced_parameters_and_shapes():
    """the per-block tensor cache of the analytic SchNet path is validated tensor by tensor (a replaced W1 / bias /
    U1 must reach the kernels and keep its own gradient slot), and the structure key names the layer shapes and whether the
    Gaussian width is trainable."""
    from mdgrad_amd.nn import analytic, get_model
    net = get_model({"n_atom_basis": 16, "n_filters": 32, "n_gaussians": 8, "n_convolutions": 2, "cutoff": 4.0})
    conv = net.convolutions[0]
    P0 = analytic._layer_params(conv)
    f = conv.moduledict["message_edge_filter"]
    assert P0["W1"] is f[1].weight and P0["c1"] is conv.moduledict["update_function"][0].bias
    assert analytic._layer_params(conv)["W1"] is P0["W1"], "unchanged modules: the cached objects"
    for mod, attr, role in ((f[1], "weight", "W1"), (f[1], "bias", "b1"), (conv.moduledict["update_function"][0], "weight", "U1"),
                            (conv.moduledict["message_node_filter"], "bias", "bn")):
        new = torch.nn.Parameter(getattr(mod, attr).detach().clone() * 2)
        setattr(mod, attr, new)
        assert analytic._layer_params(conv)[role] is new, role
    k0 = analytic._structure_key(net)
    assert analytic._structure_key(net) == k0
    f[0].width = torch.nn.Parameter(f[0].width.detach().clone())          # a trainable basis changes what `supported` reads
    assert analytic._structure_key(net) != k0
    k1 = analytic._structure_key(net)
    f[1].weight = torch.nn.Parameter(torch.zeros(8, 8))
    assert analytic._structure_key(net) == k1, "same shapes: same structure"
    f[3].weight = torch.nn.Parameter(torch.zeros(64, 8))
    assert analytic._structure_key(net) != k1, "a layer shape is part of the key"


def test_public_message_passing_names_reproduce_the_reference_energy_and_force():
    """VERDICT r5 missing #5: `MessagePassingModule` (nff/nn/graphconv.py:11-53), `scatter_add` (nff/utils/scatter.py:24-45),
    `split_and_sum` / `batch_and_sum` (nff/nn/graphop.py:9-63) are importable under the reference's names and are what
    `SchNetConv` / `SchNet.forward` run when no topology is attached: energy and `energy_grad` of golden G8 (the reference's
    own SchNet on its own list) through exactly these calls, host tensors, no kernel."""
    import numpy as np
    import torch
    from conftest import load_golden
    from mdgrad_amd import nn as mnn
    from mdgrad_amd.nn import MessagePassingModule, scatter_add, split_and_sum, batch_and_sum, compute_grad

    g = load_golden("schnet_cg64")
    net = mnn.SchNet({"n_atom_basis": int(g["n_atom_basis"]), "n_filters": int(g["n_filters"]), "n_gaussians": int(g["n_gaussians"]),
                      "n_convolutions": int(g["n_convolutions"]), "cutoff": float(g["cutoff"]), "trainable_gauss": False})
    net.load_state_dict({k[4:]: torch.as_tensor(v) for k, v in g.items() if k.startswith("sd__")})
    assert all(isinstance(c, MessagePassingModule) for c in net.convolutions)
    xyz = torch.tensor(g["pos"], dtype=torch.float32, requires_grad=True)
    nbr = torch.as_tensor(g["nbr"]).long()
    off = torch.as_tensor(g["offsets"]).float()          # (raw image flags, as GNNPotentials hands them over: SURVEY 0.8)
    batch = {"nxyz": torch.cat((torch.as_tensor(g["numbers"]).float()[:, None], xyz.detach()), 1),
             "num_atoms": torch.tensor([xyz.shape[0]]), "nbr_list": nbr, "offsets": off, "energy": None, "energy_grad": None}
    out = net(batch, xyz)
    assert set(out) == {"energy", "energy_grad"}
    assert abs(float(out["energy"].detach()) - float(g["U"][0])) <= 1e-5 + 1e-5 * abs(float(g["U"][0]))
    fmax = np.abs(g["F"]).max()
    assert np.abs(-out["energy_grad"].detach().numpy() - g["F"]).max() <= 1e-4 * fmax
    # the helpers on their own, against plain numpy
    rng = np.random.default_rng(3)
    src, idx = rng.standard_normal((7, 3)).astype(np.float32), np.array([0, 2, 2, 1, 0, 4, 2])
    want = np.zeros((5, 3), np.float32)
    np.add.at(want, idx, src)
    got = scatter_add(src=torch.as_tensor(src), index=torch.as_tensor(idx), dim=0, dim_size=5)
    assert np.allclose(got.numpy(), want, atol=1e-6)
    assert scatter_add(torch.as_tensor(src), torch.as_tensor(idx), dim=0).shape == (5, 3)        # size from the largest index
    assert np.allclose(scatter_add(torch.ones(4), torch.tensor([1, 1, 0, 1]), fill_value=2.0).numpy(), [3.0, 5.0])
    parts = split_and_sum(torch.as_tensor(src), [3, 4])
    assert np.allclose(parts.numpy(), np.stack([src[:3].sum(0), src[3:].sum(0)]), atol=1e-6)
    x = torch.tensor([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]], requires_grad=True)
    res = batch_and_sum({"energy": (x * x).sum(1, keepdim=True), "other": x}, [1, 2], ["energy_grad"], x)
    assert set(res) == {"energy", "energy_grad"} and torch.allclose(res["energy_grad"], 2 * x)
    assert torch.allclose(compute_grad(x, (x ** 3).sum()), 3 * x ** 2)
    # a user subclass written against the reference's class: messages without a filter network
    mp = MessagePassingModule()
    r, e, a = torch.ones(3, 2), torch.full((2, 2), 0.5), torch.tensor([[0, 1], [1, 2]])
    assert torch.allclose(mp(r, e, a), torch.tensor([[0.5, 0.5], [1.0, 1.0], [0.5, 0.5]]))
