"""Energy calculators with the reference's interface (torchmd/interface.py):
GeneralInteraction :33-57, PairPotentials :217-300, Stack :364-403 (GNNPotentials lives in
mdgrad_amd.nn).  forward(xyz) -> energy; _reset_topology(xyz) rebuilds the neighbour list.
"""
import inspect
import os

import torch
from torch.nn import ModuleDict

from . import _lib, ops
from .potentials import is_builtin_form
from .topology import compute_dis, get_offsets


def _accumulate_list(accum, params, grads):
    """flat[offset(p)] += weight * g for a member that returned its parameter gradients as a list."""
    jobs = ops.GradJobs()
    for p, g in zip(params, grads):
        if g is not None:
            jobs.axpy(accum.off[id(p)], g.detach().to(torch.float32))
    jobs.run(accum, alpha=1.0, accumulate=True)


class GeneralInteraction(torch.nn.Module):
    def __init__(self, system):
        super().__init__()
        self.system = system
        self.device = system.device
        self.cell = torch.Tensor(system.get_cell()).to(system.device)     # interface.py:55-56
        self.cell.requires_grad = True
        self._cell_struct = _lib.make_cell(system.get_cell())
        self._group = getattr(system, "group_size", system.get_number_of_atoms())
        self._static = None          # fixed-capacity configuration (persistent once created)
        self._static_on = False      # ... and whether _reset_topology uses it

    def _shared_ell(self, xyz, cache):
        """Exact-size ELL list at xyz; members of one Stack with the same cutoff / selection / grouping share
        it within a rebuild (`cache` is the Stack's per-call dict)."""
        key = (float(self.cutoff), None if self._mask is None else self._mask.data_ptr(), self._group)
        if cache is not None and key in cache:
            return cache[key]
        ell = ops.build_ell(xyz.detach(), self._cell_struct, self.cutoff, self._mask, group=self._group)
        if cache is not None:
            cache[key] = ell
        return ell

    def _shared_static_ell(self, xyz, cache, st):
        """Fixed-capacity ELL list (no host sync); members of one Stack with the same cutoff / selection / grouping
        and row capacity share one search per rebuild, like _shared_ell (the first member's `need` buffer carries
        the overflow report for all of them)."""
        key = ("static", float(self.cutoff), None if self._mask is None else self._mask.data_ptr(), self._group,
               int(st["max_nbr"]))
        if cache is not None and key in cache:
            return cache[key]
        ell = ops.build_ell(xyz.detach(), self._cell_struct, self.cutoff, self._mask, max_nbr=st["max_nbr"],
                            group=self._group, need=st["need"])
        if cache is not None:
            cache[key] = ell
        return ell

    # Verlet reuse of the fixed-capacity lists (graph replay / sync-free eager passes only): the list is searched with
    # cutoff * (1 + verlet_skin) and kept while no atom has moved more than half the skin; consumers re-apply the exact
    # cutoff, so every evaluation sees the pair set of a fresh search (ops.VerletList).  0 switches it off.
    verlet_skin = float(os.environ.get("MDG_VERLET_SKIN", "0.02"))

    def _verlet_list(self, xyz, cache, st):
        key = ("verlet", float(self.cutoff), None if self._mask is None else self._mask.data_ptr(), self._group)
        if cache is not None and key in cache:
            return cache[key]
        vl = st.get("verlet")
        skin = self.verlet_skin * float(self.cutoff)
        # everything the stored list was searched with: a changed cutoff / skin / selection / cell / grouping rebuilds it
        sig = (int(st["max_nbr"]), int(st["capacity"]), float(self.cutoff), float(skin),
               None if self._mask is None else self._mask.data_ptr(), bytes(self._cell_struct), int(self._group))
        if vl is None or vl.sig != sig or vl.n_atoms != xyz.shape[0]:
            vl = st["verlet"] = ops.VerletList(xyz.shape[0], self._group, self._cell_struct, self.cutoff,
                                               skin, self._mask, sig[0], sig[1], xyz.device)
            vl.sig = sig
        vl.rebuild(xyz, st["need"])
        if cache is not None:
            cache[key] = vl
        return vl

    # -- fixed-capacity neighbour lists (HIP-graph capture of the integrator steps, mdgrad_amd/graphs.py) --
    def supports_static_topology(self):
        return False

    def static_overflow(self):
        """True when a fixed-capacity rebuild since the last check did not fit (one host sync); the
        capacities are enlarged so that the caller can redo the pass."""
        st = self._static
        if st is None:
            return False
        need = st["need"].tolist()
        if need[0] <= st["max_nbr"] and need[1] <= st.get("capacity", 1 << 62):
            return False
        if need[0] > st["max_nbr"]:
            st["max_nbr"] = min(self._group - 1, (int(need[0] * 1.25) + 15) // 8 * 8)
        if need[1] > st.get("capacity", 1 << 62):
            st["capacity"] = (int(need[1] * 1.25) + 1023) // 1024 * 1024
        st["need"].zero_()
        st["version"] += 1
        return True

    def static_version(self):
        return -1 if self._static is None else self._static["version"]


class _LazyTopology:
    """(nbr_list, pair_dis, offsets) of PairPotentials._reset_topology, materialised on demand
    (the integrators ignore the return value; materialising costs a host sync)."""

    def __init__(self, owner, xyz):
        self._owner, self._xyz, self._val = owner, xyz, None

    def _get(self):
        if self._val is None:
            nbr, off = self._owner._ell.half_list()
            dis = compute_dis(self._xyz, nbr, off, self._owner.cell.detach()).reshape(-1)
            self._val = (nbr, dis, off)
        return self._val

    def __iter__(self):
        return iter(self._get())

    def __getitem__(self, k):
        return self._get()[k]

    def __len__(self):
        return 3


def batch_to(batch, device):
    """nff/utils/cuda.py:6-10."""
    return {k: (v.to(device) if hasattr(v, 'to') else v) for k, v in batch.items()}


class GNNPotentials(GeneralInteraction):
    """torchmd/interface.py:86-136.  `inputs` is the batch dict the SchNet consumes
    ('nxyz', 'num_atoms', 'energy', 'nbr_list', 'offsets'); `_reset_topology` rebuilds the list
    with the HIP builder and also stores the per-atom topology the cfconv kernels use."""

    def __init__(self, system, gnn, cutoff, ex_pairs=None):
        super().__init__(system)
        self.gnn = gnn
        self.cutoff = cutoff
        self.inputs = batch_to(self.system.get_batch(), self.device)
        self.inputs['cell'] = self.cell.detach()
        self.ex_pairs = ex_pairs
        self._mask = ops.build_mask(self._group, None, ex_pairs, system.device)
        self.to(self.device)
        self._reset_topology(torch.Tensor(system.get_positions()).to(system.device))

    def _reset_topology(self, xyz, _cache=None):
        self._topo_stamp = object()                  # identity of this rebuild (see md._EOM.update_topology)
        st = self._static if self._static_on else None
        if st is not None and self.verlet_skin > 0 and self._verlet_consumers_ok():
            topo = self._verlet_list(xyz, _cache, st).topo               # (rebuilt only when an atom left its half-skin ball)
        elif st is not None:
            topo = ops.StaticTopo(self._shared_static_ell(xyz, _cache, st), st["capacity"], st["need"])
        else:
            topo = ops.GraphTopo(self._shared_ell(xyz, _cache))
        self.inputs['nbr_list'], self.inputs['offsets'] = topo.nbr, topo.offsets
        self.inputs['_topo'] = topo

    def supports_static_topology(self):
        return self.supports_force_vjp()

    def _verlet_consumers_ok(self):
        """A list searched with a skin is only handed to consumers that re-apply the exact cutoff per pair: the fused
        interaction-block kernels (`ops.edge_geom` -> mdg_edge_geom_masked marks pairs beyond the cutoff, the cfconv
        kernels skip them).  The unfused chain (`analytic._primal`: n_gaussians > 64, `fused_block = False`, filter
        counts the fused kernels do not take) reads every listed pair at full weight -- the filter has no cutoff
        envelope -- so it gets the exact fixed-capacity list."""
        from .nn import analytic
        return self.supports_force_vjp() and analytic.fused_ok(self.gnn)

    def set_static_topology(self, on=True):
        """Fixed capacities (neighbours per atom, edges) sized from the current list with ~25 % head room."""
        self._static_on = bool(on)
        if on and self._static is None:
            topo = self.inputs['_topo']
            longest = int(topo.ell.cnt.max().item())
            grow = 1.25 * (1.0 + self.verlet_skin) ** 3              # head room, and the skin's extra candidates
            self._static = dict(max_nbr=min(self._group - 1, (int(longest * grow) + 15) // 8 * 8),
                                capacity=(int(topo.n_edges * grow) + 1023) // 1024 * 1024,
                                need=torch.zeros(2, dtype=torch.int32, device=self.device), version=0)

    def forward(self, xyz):
        results = self.gnn(self.inputs, xyz)
        return results['energy']

    # -- analytic-adjoint protocol: hand-derived SchNet passes (mdgrad_amd/nn/analytic.py) ---------
    analytic = True

    def supports_force_vjp(self):
        from .nn import analytic
        return self.analytic and analytic.supported(self.gnn) and not getattr(self.gnn, "cartesian_offsets", False)

    def _z(self):
        z = getattr(self, "_z_cache", None)
        if z is None or z.shape[0] != self.inputs['nxyz'].shape[0]:
            z = self._z_cache = self.inputs['nxyz'][:, 0].long()
        return z

    def force(self, xyz):
        from .nn import analytic
        return analytic.force(self.gnn, self._z(), xyz, self.inputs['_topo'], self.inputs['offsets'], want_energy=False)[1]

    accepts_accum = True

    def prepare_pass(self):
        """Once per trajectory pass, before HIP-graph replays: state the captured steps read but do not recompute (the
        embedding rows of the atoms, which change with every optimizer step)."""
        from .nn import analytic
        if self.supports_force_vjp():
            analytic.refresh_embedding(self.gnn, self._z())

    def force_vjp(self, xyz, w, want_theta=True, accum=None):
        """`accum` (ops.ThetaAccum): the parameter gradients are added into its flat buffer (weighted on the device) and
        None is returned in their place."""
        from .nn import analytic
        _, F, dq, gth = analytic.force_vjp(self.gnn, self._z(), xyz, w, self.inputs['_topo'], self.inputs['offsets'],
                                           want_theta=want_theta, want_energy=False, accum=accum)
        return F, dq, gth


class PairPotentials(GeneralInteraction):
    """torchmd/interface.py:217-300.  `pair_model` is a module instance (the code's signature)
    or, as in the README snippet, a class plus its keyword arguments."""

    def __init__(self, system, pair_model, cutoff=2.5, index_tuple=None, ex_pairs=None,
                 nbr_list_device=None, **model_kwargs):
        super().__init__(system)
        if inspect.isclass(pair_model):
            pair_model = pair_model(**model_kwargs)
        elif model_kwargs:
            raise TypeError("unexpected keyword arguments %s" % list(model_kwargs))
        self.nbr_list_device = system.device if nbr_list_device is None else nbr_list_device
        self.model = pair_model.to(system.device)
        self.cutoff = cutoff
        self.index_tuple = index_tuple
        self.ex_pairs = ex_pairs
        self._mask = ops.build_mask(self._group, index_tuple, ex_pairs, system.device)
        self._nbr_override = None
        self._reset_topology(torch.Tensor(system.get_positions()).to(system.device))

    # -- kernel-facing description ------------------------------------------------------
    def builtin(self):
        return is_builtin_form(self.model)

    def mdg_term(self, theta_off=0):
        params = self.model.mdg_params()
        return ops.make_term(self.model.mdg_term(), self.cutoff, theta_off,
                             sum(p.numel() for p in params), self._mask)

    # -- reference attributes -----------------------------------------------------------
    @property
    def nbr_list(self):
        """[P,2] int64 half list.  (The reference parks it on the CPU, interface.py:259; it
        stays on the device here.)"""
        return self._ell.half_list()[0]

    @property
    def offsets(self):
        return self._ell.half_list()[1]

    def _reset_topology(self, xyz, _cache=None):
        self._topo_stamp = object()
        st = self._static if self._static_on else None
        vkey = ("verlet", float(self.cutoff), None if self._mask is None else self._mask.data_ptr(), self._group)
        if st is not None and _cache is not None and vkey in _cache and self.builtin():
            self._ell = _cache[vkey].ell             # a Stack member's Verlet list (exact cutoff re-applied per pair)
        elif st is not None:
            self._ell = self._shared_static_ell(xyz, _cache, st)
        else:
            self._ell = self._shared_ell(xyz, _cache)
        return _LazyTopology(self, xyz.detach())

    def supports_static_topology(self):
        # module pair models differentiate phi(r) with the autograd engine, whose worker thread cannot
        # take part in a stream capture here (segfault at capture end on ROCm 7): eager launches for them
        return self.builtin()

    def set_static_topology(self, on=True):
        self._static_on = bool(on)
        if on and self._static is None:
            longest = int(self._ell.cnt.max().item())
            self._static = dict(max_nbr=min(self._group - 1, (int(longest * 1.25) + 15) // 8 * 8),
                                need=torch.zeros(2, dtype=torch.int32, device=self.device), version=0)
            if not self.builtin():                    # module path works on the half list: edge capacity too
                pairs = int(self._ell.half_list()[0].shape[0])
                self._static["capacity"] = (int(pairs * 1.25) + 1023) // 1024 * 1024

    # -- analytic-adjoint protocol: used by the integrators' rhs_vjp --------------------------------
    analytic = True            # False: user-defined pair modules go through the autograd double-backward path

    def supports_force_vjp(self):
        return self.builtin() or self.analytic

    # user-defined pair module phi(r) (pairMLP ...): phi', phi'' and the parameter vjp by autograd over the
    # [P] pair distances only; geometry and the per-atom sums on the HIP list (no float atomics)
    def _phi(self, r):
        return self.model(r)

    def _module_pairs(self, xyz):
        st = self._static if self._static_on else None
        topo = ops.GraphTopo(self._ell) if st is None else ops.StaticTopo(self._ell, st["capacity"], st["need"])
        cellm = self.cell.detach()
        cellm = torch.diag(cellm) if cellm.dim() == 1 else cellm
        delta = ops._edge_diff(xyz, topo) - topo.offsets.matmul(cellm)        # compute_dis, topology.py:9-10
        d = delta.pow(2).sum(1).sqrt()
        return topo, delta / d[:, None], d

    def _module_force(self, xyz):
        topo, uhat, d = self._module_pairs(xyz)
        with torch.enable_grad():
            r = d.detach().requires_grad_(True)
            (g1,) = torch.autograd.grad(self._phi(r[:, None]).sum(), r)
        return -ops._edge_scatter(g1[:, None] * uhat, topo)

    def _module_force_vjp(self, xyz, w, want_theta):
        topo, uhat, d = self._module_pairs(xyz)
        ddel = ops._edge_diff(w, topo)
        a = (uhat * ddel).sum(1)                                             # rhat . (w_i - w_j)
        params = [p for p in self.model.parameters() if p.requires_grad] if want_theta else []
        with torch.enable_grad():
            r = d.detach().requires_grad_(True)
            (g1,) = torch.autograd.grad(self._phi(r[:, None]).sum(), r, create_graph=True)
            # S = w . grad U = sum_p phi'(r_p) a_p :  dS/dr_p = phi''(r_p) a_p ,  dS/dtheta = d(w . grad U)/dtheta
            grads = torch.autograd.grad((g1 * a).sum(), [r] + params, allow_unused=True)
        g1 = g1.detach()
        g2a = grads[0]
        hv = g2a[:, None] * uhat + (g1 / d)[:, None] * (ddel - a[:, None] * uhat)
        F = -ops._edge_scatter(g1[:, None] * uhat, topo)
        by_id = {id(p): g_ for p, g_ in zip(params, grads[1:])}
        gth = [(-by_id[id(p)] if by_id.get(id(p)) is not None else torch.zeros_like(p))
               for p in self.model.parameters()] if want_theta else None
        return F, -ops._edge_scatter(hv, topo), gth

    def _theta(self, like):
        """(flat parameter vector of the pair form, its parameters).  The vector lives in a persistent buffer that is
        refreshed when a parameter changed (version counters); inside a HIP-graph capture it is returned as it is -- the
        replaying pass refreshes it once before its first replay (`prepare_pass`), so the captured steps carry no `cat`."""
        params = self.model.mdg_params()
        if not params:
            return like.new_zeros(0), params
        n = sum(p.numel() for p in params)
        buf = getattr(self, "_theta_buf", None)
        if buf is None or buf.numel() != n or buf.device != params[0].device:
            if params[0].is_cuda and torch.cuda.is_current_stream_capturing():
                return torch.cat([p.detach().reshape(-1) for p in params]), params
            buf = self._theta_buf = torch.empty(n, device=params[0].device, dtype=torch.float32)
            self._theta_key = None
        if params[0].is_cuda and torch.cuda.is_current_stream_capturing():
            return buf, params
        key = tuple((p.data_ptr(), p._version) for p in params)
        if self._theta_key != key:
            torch.cat([p.detach().reshape(-1).to(torch.float32) for p in params], out=buf)
            self._theta_key = key
        return buf, params

    def prepare_pass(self):
        if self.builtin():
            self._theta(self.cell)

    accepts_into = True

    def force(self, xyz, into=None):
        """F = -dU/dx in one kernel launch; `into` (a force buffer of another Stack member): added onto it in the same
        launch and returned."""
        if not self.builtin():
            f = self._module_force(xyz.detach().contiguous())
            return f if into is None else into.add_(f)
        theta, _ = self._theta(xyz)
        o = ops.pair_eval(self._ell, xyz.detach().contiguous(), self.mdg_term(0), theta, energy=False, grad=True,
                          into=None if into is None else (into, None), scale=-1.0, theta_grads=False)
        return o["grad"]

    accepts_accum = True

    def force_vjp(self, xyz, w, want_theta=True, accum=None, into=None):
        """(F, d(w.F)/dx, [d(w.F)/dtheta_p for p in parameters()]) -- what double autograd yields at
        torchmd/sovlers.py:229-233 -- in one kernel launch (force + Hessian-vector product + mixed term).  `accum`
        (ops.ThetaAccum): the parameter part is added into its flat buffer instead (None returned).  `into` = (F, dq)
        buffers of another Stack member: this term's force and d(w.F)/dx are added onto them in the same launch."""
        if not self.builtin():
            out = self._module_force_vjp(xyz.detach().contiguous(), w.detach().contiguous(), want_theta)
            if into is not None:
                out = (into[0].add_(out[0]), into[1].add_(out[1]), out[2])
            if accum is None or out[2] is None:
                return out
            _accumulate_list(accum, self.parameters(), out[2])
            return out[0], out[1], None
        theta, params = self._theta(xyz)
        o = ops.pair_eval(self._ell, xyz.detach().contiguous(), self.mdg_term(0), theta, w=w.detach().contiguous(),
                          energy=False, grad=True, into=into, scale=-1.0, theta_grads=bool(want_theta))
        if accum is not None and want_theta:
            if params:
                jobs, pos = ops.GradJobs(), 0
                offs = [accum.off[id(p)] for p in params]
                if all(offs[k + 1] == offs[k] + params[k].numel() for k in range(len(params) - 1)):
                    jobs.axpy(offs[0], o["gtheta_w"])               # the term's parameters are adjacent in the flat buffer
                else:
                    for p, o_ in zip(params, offs):
                        jobs.axpy(o_, o["gtheta_w"][pos:pos + p.numel()])
                        pos += p.numel()
                jobs.run(accum, alpha=-1.0, accumulate=True)
            return o["grad"], o["hw"], None
        if not want_theta:
            return o["grad"], o["hw"], None
        gth, pos = [], 0
        for p in params:
            n = p.numel()
            gth.append(-o["gtheta_w"][pos:pos + n].reshape(p.shape))
            pos += n
        return o["grad"], o["hw"], gth

    def forward(self, xyz):
        if self.builtin():
            params = self.model.mdg_params()
            theta = (torch.cat([p.reshape(-1) for p in params]) if params
                     else xyz.new_zeros(0))
            return ops.PairEnergyFn.apply(xyz.contiguous(), theta, self._ell, self.mdg_term(0))
        # user-defined pair module (e.g. an MLP): distances with torch ops on the device
        nbr, off = self._ell.half_list()
        pair_dis = compute_dis(xyz, nbr, off, self.cell)
        return self.model(pair_dis).sum()


class TPairPotentials(PairPotentials):
    """Temperature-dependent pair model u(r, kB T) (torchmd/interface.py:139-215; e.g. potentials.TpairMLP)."""

    def __init__(self, system, pair_model, T, cutoff=2.5, index_tuple=None, ex_pairs=None, nbr_list_device=None):
        super().__init__(system, pair_model, cutoff=cutoff, index_tuple=index_tuple, ex_pairs=ex_pairs,
                         nbr_list_device=nbr_list_device)
        self.T = T

    def builtin(self):
        return False

    def _phi(self, r):
        from . import units
        return self.model(r, units.kB * self.T)

    def forward(self, xyz):
        nbr, off = self._ell.half_list()
        return self._phi(compute_dis(xyz, nbr, off, self.cell)).sum()            # interface.py:207-215


class _BondedTerm(torch.nn.Module):
    """Common part of BondPotentials / AnglePotentials: the static topology table on the device and the analytic-adjoint
    protocol (force / force_vjp) on mdg_bonded_eval (csrc/bonded.hip) -- one launch for energy, force and the
    Hessian-vector product, so a polymer Stack(pair + bond [+ angle] [+ GNN]) (demo/fold.py:131-161) keeps the analytic
    adjoint and HIP-graph replay instead of the reference's double backward.  No trainable parameters (k, ro / thetao are
    plain numbers in the reference); given as tensors that require grad they switch the term to the reference's torch
    ops so that autograd reaches them."""

    _kind = None
    accepts_into = True
    accepts_accum = True
    analytic = True

    def __init__(self, system, top):
        super().__init__()
        self.system = system
        self.device = system.device
        self.cell = torch.Tensor(system.get_cell()).diag().to(self.device)      # interface.py:429-431
        self.top = top.to(self.device)
        self._n_atoms = system.get_number_of_atoms()
        self._table = None

    def _consts(self):
        raise NotImplementedError

    def _hip_ok(self, xyz=None):
        """The HIP kernel serves float32 device positions with plain-number constants.  Constants that require grad, and
        float64 device positions, take the reference's torch ops on the device instead; host tensors are refused
        by the kernel entry as everywhere (no CPU path)."""
        if xyz is not None and xyz.is_cuda and xyz.dtype != torch.float32:
            return False
        return self.analytic and not any(torch.is_tensor(c) and c.requires_grad for c in self._consts())

    def table(self):
        k, x0 = [float(c) for c in self._consts()]
        t = self._table
        if t is None or t.k != k or t.x0 != x0:
            t = self._table = ops.BondedTable(self._kind, self.top, self._n_atoms, self.cell.detach().cpu().tolist(), k, x0,
                                              self.device)
        return t

    def _reset_topology(self, xyz):          # (static table; absent for AnglePotentials in the reference, which therefore
        pass                                 #  cannot be a member of a Stack there)

    def _torch_energy(self, xyz):
        raise NotImplementedError

    def forward(self, xyz):
        if self._hip_ok(xyz):
            return ops.BondedEnergyFn.apply(xyz.contiguous(), self.table())
        return self._torch_energy(xyz)

    # -- analytic-adjoint protocol (md._EOM.rhs_vjp, Stack.force / force_vjp) -------------------------------------
    def supports_force_vjp(self):
        return self._hip_ok()

    def force(self, xyz, into=None):
        o = ops.bonded_eval(self.table(), xyz.detach(), into=None if into is None else (into, None), scale=-1.0)
        return o["grad"]

    def force_vjp(self, xyz, w, want_theta=True, accum=None, into=None):
        """(F, d(w.F)/dx, []) -- the term has no parameters."""
        o = ops.bonded_eval(self.table(), xyz.detach(), w=w.detach(), into=into, scale=-1.0)
        return o["grad"], o["hw"], ([] if (want_theta and accum is None) else None)

    # -- fixed-capacity topology (HIP-graph capture): the table is static, nothing can overflow ------------------
    def supports_static_topology(self):
        return self._hip_ok()

    def set_static_topology(self, on=True):
        self.table()

    def static_overflow(self):
        return False

    def static_version(self):
        return 0


class BondPotentials(_BondedTerm):
    """Harmonic term in the SQUARED bond length, 1/2 k (|b|^2 - ro)^2 (torchmd/interface.py:406-456; the polymer demo's
    bonded term, demo/fold.py:131).  `top` = [n_bonds, 2] atom indices; orthorhombic minimum image with the non-strict
    test of topology.get_offsets (topology.py:75-80)."""

    _kind = _lib.BONDED_BOND

    def __init__(self, system, top, k, ro):
        super().__init__(system, top)
        self.k, self.ro = k, ro

    def _consts(self):
        return self.k, self.ro

    def _torch_energy(self, xyz):
        b = xyz[self.top[:, 0]] - xyz[self.top[:, 1]]
        b = b + get_offsets(b, self.cell, self.device) * self.cell
        return 0.5 * self.k * (b.pow(2).sum(-1) - self.ro).pow(2).sum(-1)


class AnglePotentials(_BondedTerm):
    """Harmonic angle term 1/2 k (theta - theta0)^2 over triples (i, j, k) centred on j
    (torchmd/interface.py:457-510)."""

    _kind = _lib.BONDED_ANGLE

    def __init__(self, system, top, k, thetao):
        super().__init__(system, top)
        self.k, self.thetao = k, thetao

    def _consts(self):
        return self.k, self.thetao

    def _torch_energy(self, xyz):
        b1 = xyz[self.top[:, 0]] - xyz[self.top[:, 1]]
        b2 = xyz[self.top[:, 2]] - xyz[self.top[:, 1]]
        b1 = b1 + get_offsets(b1, self.cell, self.device) * self.cell
        b2 = b2 + get_offsets(b2, self.cell, self.device) * self.cell
        cos = (b1 * b2).sum(-1) / (b1.pow(2).sum(-1) * b2.pow(2).sum(-1)).sqrt()
        return 0.5 * self.k * (torch.acos(cos) - self.thetao).pow(2).sum(-1)


class Stack(torch.nn.Module):
    """torchmd/interface.py:364-403."""

    def __init__(self, model_dict, mode='sum'):
        super().__init__()
        self.models = ModuleDict(model_dict)

    # analytic-adjoint protocol: available when every member provides it
    def supports_force_vjp(self):
        return all(getattr(m, "supports_force_vjp", lambda: False)() for m in self.models.values())

    def prepare_pass(self):
        for m in self.models.values():
            getattr(m, "prepare_pass", lambda: None)()

    def _ordered(self):
        """Members that can add their result onto an existing buffer inside their own launch (pair terms) go last."""
        ms = list(self.models.values())
        return [m for m in ms if not getattr(m, "accepts_into", False)] + [m for m in ms if getattr(m, "accepts_into", False)]

    def force(self, x):
        out = None
        for m in self._ordered():
            if out is not None and getattr(m, "accepts_into", False):
                out = m.force(x, into=out)
                continue
            f = m.force(x)
            out = f if out is None else out + f
        return out

    accepts_accum = True

    def force_vjp(self, x, w, want_theta=True, accum=None):
        """Sum over members; the parameter gradients come back as a list aligned with self.parameters()
        (None when want_theta is False: the first augmented evaluation of an adjoint interval discards
        them, sovlers.py:141-143) -- or are added into `accum` (ops.ThetaAccum) by the members themselves."""
        F = dq = None
        by_id = {}
        for m in self._ordered():
            kw = {}
            if F is not None and getattr(m, "accepts_into", False):
                kw["into"] = (F, dq)                                   # this member's launch adds onto the running sums
            if accum is not None and want_theta:
                if getattr(m, "accepts_accum", False):
                    f, g, gth = m.force_vjp(x, w, want_theta=True, accum=accum, **kw)
                else:
                    f, g, gth = m.force_vjp(x, w, want_theta=True, **kw)
                if gth is not None:
                    _accumulate_list(accum, m.parameters(), gth)
            else:
                f, g, gth = m.force_vjp(x, w, want_theta=want_theta, **kw)
            if "into" in kw:
                F, dq = f, g
            else:
                F = f if F is None else F + f
                dq = g if dq is None else dq + g
            if want_theta and accum is None:
                for p, gp in zip(m.parameters(), gth):
                    by_id[id(p)] = gp if id(p) not in by_id else by_id[id(p)] + gp
        if not want_theta or accum is not None:
            return F, dq, None
        return F, dq, [by_id[id(p)] if id(p) in by_id else torch.zeros_like(p) for p in self.parameters()]

    def _reset_topology(self, x):
        self._topo_stamp = object()
        shared = {}                          # one neighbour search per distinct (cutoff, selection, grouping)
        # (members that own a stored Verlet list go first, so that pair terms with the same cutoff can use it)
        keys = sorted(self.models.keys(), key=lambda k: 0 if isinstance(self.models[k], GNNPotentials) else 1)
        for key in keys:
            m = self.models[key]
            if isinstance(m, GeneralInteraction):
                m._reset_topology(x, _cache=shared)
            else:
                m._reset_topology(x)

    def supports_static_topology(self):
        return all(getattr(m, "supports_static_topology", lambda: False)() for m in self.models.values())

    def set_static_topology(self, on=True):
        for m in self.models.values():
            m.set_static_topology(on)

    def static_overflow(self):
        return any([m.static_overflow() for m in self.models.values()])     # (no short-circuit: all get enlarged)

    def static_version(self):
        return tuple(m.static_version() for m in self.models.values())

    def forward(self, x):
        result = None
        for key in self.models.keys():
            new_result = self.models[key](x).sum().reshape(-1)
            result = new_result if result is None else result + new_result
        return result
