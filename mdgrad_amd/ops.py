"""Device ops: thin Python over the C ABI (include/mdgrad_hip.h) plus the
torch.autograd.Function wrappers that own the differentiation contract

    PairEnergyFn -> PairGradFn -> (Hessian-vector product)     pair potentials, any order <= 2
    FusedTrajFn                                                 whole NH-Verlet/Verlet trajectory + adjoint
    RdfRawFn                                                    soft histogram

Everything here requires HIP tensors; there is no CPU path.
"""
import ctypes as C
import math
import weakref

import torch
from array import array as _array

from . import _lib
from . import _torch_ops
from ._lib import MdgPairTerm, MdgTerms, MdgTrajParams, check, ptr, stream_ptr, require_gpu


# ----------------------------------------------------------------------------- neighbour lists
def build_mask(n_atoms, index_tuple=None, ex_pairs=None, device=None):
    """[N,N] uint8 selection equivalent to the reference's multiplicative masks
    (torchmd/topology.py:15-27 index_tuple product, :44-53 ex_pairs).  None when unrestricted."""
    if index_tuple is None and ex_pairs is None:
        return None
    keep = torch.ones(n_atoms, n_atoms, dtype=torch.bool)
    if index_tuple is not None:
        a = torch.as_tensor(list(index_tuple[0]), dtype=torch.long)
        b = torch.as_tensor(list(index_tuple[1]), dtype=torch.long)
        sel = torch.zeros(n_atoms, n_atoms, dtype=torch.bool)
        sel[a[:, None], b[None, :]] = True
        keep &= sel | sel.t()
    if ex_pairs is not None:
        ex = torch.as_tensor(ex_pairs, dtype=torch.long).reshape(-1, 2).cpu()
        keep[ex[:, 0], ex[:, 1]] = False
        keep[ex[:, 1], ex[:, 0]] = False
    return keep.to(torch.uint8).contiguous().to(device)


def estimate_max_nbr(n_atoms, cell_struct, cutoff):
    h = cell_struct.h
    vol = abs(h[0] * (h[4] * h[8] - h[5] * h[7]) - h[1] * (h[3] * h[8] - h[5] * h[6])
              + h[2] * (h[3] * h[7] - h[4] * h[6]))
    est = n_atoms / max(vol, 1e-30) * 4.0 / 3.0 * math.pi * cutoff ** 3
    cap = int(est * 1.5) + 16
    cap = (cap + 7) // 8 * 8
    return max(8, min(max(n_atoms - 1, 1), cap))


class EllList:
    """Immutable device snapshot of one neighbour list (layout: include/mdgrad_hip.h K1)."""

    def __init__(self, n_atoms, max_nbr, col, shift, cnt, cell_struct, cutoff, mask):
        self.n_atoms, self.max_nbr = n_atoms, max_nbr
        self.col, self.shift, self.cnt = col, shift, cnt
        self.cell_struct, self.cutoff, self.mask = cell_struct, cutoff, mask
        self._half = None

    def half_list(self, with_edge_id=False):
        """(nbr int64 [P,2], offsets f32 [P,3][, edge_id int32 [N,max_nbr]]) in the reference's
        lexicographic i<j order (torchmd/topology.py:68-73).  One host sync (P)."""
        if self._half is None or (with_edge_id and self._half[2] is None):
            lib = _lib.load()
            dev = self.col.device
            row_base = torch.empty(self.n_atoms + 1, dtype=torch.int32, device=dev)
            st = stream_ptr(dev)
            check(lib.mdg_nbr_half_count(ptr(self.col), ptr(self.cnt), self.n_atoms, self.max_nbr,
                                         ptr(row_base), st), "mdg_nbr_half_count")
            P = int(row_base[-1].item())
            nbr = torch.empty(P, 2, dtype=torch.int64, device=dev)
            off = torch.empty(P, 3, dtype=torch.float32, device=dev)
            eid = (torch.zeros(self.n_atoms, self.max_nbr, dtype=torch.int32, device=dev)
                   if with_edge_id else None)
            check(lib.mdg_nbr_half_fill(ptr(self.col), ptr(self.shift), ptr(self.cnt), ptr(row_base),
                                        self.n_atoms, self.max_nbr, ptr(nbr), ptr(off), ptr(eid), st),
                  "mdg_nbr_half_fill")
            self._half = (nbr, off, eid)
        return self._half if with_edge_id else self._half[:2]

    def half_list_padded(self, capacity, pad_offset, need):
        """Fixed-capacity half list for HIP-graph capture (no host sync): (nbr [capacity,2] with (-1,-1)
        padding rows, offsets [capacity,3] with (pad_offset,0,0) there, edge_id, n_valid int32[1]).
        need: persistent int32 buffer, need[1] <- max(need[1], pairs) when the capacity is exceeded."""
        lib = _lib.load()
        dev = self.col.device
        row_base = torch.empty(self.n_atoms + 1, dtype=torch.int32, device=dev)
        st = stream_ptr(dev)
        check(lib.mdg_nbr_half_count(ptr(self.col), ptr(self.cnt), self.n_atoms, self.max_nbr, ptr(row_base), st),
              "mdg_nbr_half_count")
        nbr = torch.empty(capacity, 2, dtype=torch.int64, device=dev)
        off = torch.empty(capacity, 3, dtype=torch.float32, device=dev)
        eid = torch.zeros(self.n_atoms, self.max_nbr, dtype=torch.int32, device=dev)
        n_valid = torch.empty(1, dtype=torch.int32, device=dev)
        check(lib.mdg_nbr_half_fill_padded(ptr(self.col), ptr(self.shift), ptr(self.cnt), ptr(row_base),
                                           self.n_atoms, self.max_nbr, int(capacity), float(pad_offset), ptr(nbr),
                                           ptr(off), ptr(eid), ptr(n_valid), C.c_void_p(need.data_ptr() + 4), st),
              "mdg_nbr_half_fill_padded")
        return nbr, off, eid, n_valid


def _use_cell_list(n_atoms, cs, cutoff):
    if not cs.diag or n_atoms < 512:
        return False
    return all(cs.h[4 * d] / cutoff >= 3.0 for d in range(3))


def build_ell(xyz, cell_struct, cutoff, mask=None, max_nbr=None, method="auto", group=None, need=None):
    """Neighbour list of one frame xyz[N,3] (replaces generate_nbr_list).  method: auto|dense|cell.
    group: atoms per independent replica when xyz stacks several replicas of one system (pairs stay
    inside a group; mask is [group, group]).
    need: persistent int32[2] buffer => fixed-capacity mode for HIP-graph capture: one pass with the
    given max_nbr, no host sync; need[0] <- max(need[0], longest row) when a row does not fit."""
    require_gpu(xyz, "xyz")
    lib = _lib.load()
    xyz = xyz.detach().contiguous()
    N, dev = xyz.shape[0], xyz.device
    cutoff = float(cutoff)
    grouped = group is not None and group != N
    Ng = group if grouped else N
    cap = estimate_max_nbr(Ng, cell_struct, cutoff) if max_nbr is None else int(max_nbr)
    use_cell = method == "cell" or (method == "auto" and _use_cell_list(Ng, cell_struct, cutoff))
    st = stream_ptr(dev)
    while True:
        if need is not None and max_nbr is None:
            raise ValueError("build_ell: fixed-capacity mode needs max_nbr")
        col = torch.empty(N, cap, dtype=torch.int32, device=dev)
        shift = torch.empty(N, cap, dtype=torch.int32, device=dev)
        cnt = torch.empty(N, dtype=torch.int32, device=dev)
        overflow = torch.zeros(1, dtype=torch.int32, device=dev) if need is None else need
        if use_cell and cap <= 512:                 # (the cell-list kernel sorts rows of up to 512 entries in LDS)
            ns = lib.mdg_nbr_cell_scratch_groups(N, Ng, C.byref(cell_struct), cutoff)
            scratch = torch.empty(int(ns), dtype=torch.int32, device=dev)
            check(lib.mdg_nbr_build_cell_groups(ptr(xyz), N, Ng, C.byref(cell_struct), cutoff, ptr(mask), ptr(col),
                                                ptr(shift), ptr(cnt), cap, ptr(overflow), ptr(scratch), st),
                  "mdg_nbr_build_cell")
        else:
            check(lib.mdg_nbr_build_dense_groups(ptr(xyz), N, Ng, C.byref(cell_struct), cutoff, ptr(mask), ptr(col),
                                                 ptr(shift), ptr(cnt), cap, ptr(overflow), st),
                  "mdg_nbr_build_dense")
        if cap >= Ng - 1 or need is not None:
            break                                   # cannot overflow / checked later by the caller
        longest = int(overflow.item())              # longest row the builder met (one host sync)
        if longest <= cap:
            break
        cap = min(Ng - 1, (longest + 15) // 8 * 8)  # grow to fit and rebuild
    return EllList(N, cap, col, shift, cnt, cell_struct, cutoff, mask)


# ----------------------------------------------------------------------------- pair potentials
MDG_PAIR_TABLE = 4


def make_term(desc, cutoff, theta_off=0, n_theta=0, mask=None):
    t = MdgPairTerm()
    t.kind = desc["kind"]
    t.p, t.q, t.c = desc.get("p", 0), desc.get("q", 0), desc.get("c", 0.0)
    t.a, t.phi = desc.get("a", 0.0), desc.get("phi", 0.0)
    t.cutoff = float(cutoff)
    t.theta_off, t.n_theta = theta_off, n_theta
    t.mask = mask.data_ptr() if mask is not None else None
    return t


def pair_eval(ell, xyz, term, theta, w=None, energy=True, grad=True, into=None, scale=1.0, theta_grads=True):
    """One launch of mdg_pair_eval_ell.  Returns dict with the requested outputs.  `into` = (grad_buffer, hw_buffer or
    None): the per-atom outputs are ADDED onto those buffers, times `scale` (F += -dU/dx of a Stack member without extra
    launches); without `into`, `scale` multiplies the fresh outputs.  theta_grads=False: dU/dtheta and d(w.dU/dx)/dtheta are
    not wanted (a force-only evaluation then skips the scalar reduction launch)."""
    lib = _lib.load()
    require_gpu(xyz, "xyz")
    if theta is not None and theta.numel():
        require_gpu(theta, "theta")
    if w is not None:
        require_gpu(w, "w")
    if xyz.shape != (ell.n_atoms, 3):
        raise ValueError("mdgrad_amd: xyz must be [%d, 3] (got %s)" % (ell.n_atoms, tuple(xyz.shape)))
    dev = xyz.device
    N, K = ell.n_atoms, term.n_theta
    out = {}
    e = torch.empty(1, device=dev) if energy else None
    acc = into is not None
    g = (into[0] if acc else torch.empty(N, 3, device=dev)) if grad else None
    gth = torch.empty(K, device=dev) if (grad and K and theta_grads) else None
    hw = (into[1] if acc else torch.empty(N, 3, device=dev)) if w is not None else None
    gthw = torch.empty(K, device=dev) if (w is not None and K and theta_grads) else None
    partial = torch.empty(int(lib.mdg_pair_partial_size(N)), device=dev)
    recheck = bool(getattr(ell, "verlet", False))        # a list searched with a skin: exact cutoff test per pair
    if acc or scale != 1.0 or recheck:
        check(lib.mdg_pair_eval_ell_into(ptr(xyz), N, C.byref(ell.cell_struct), ptr(ell.col), ptr(ell.shift),
                                         ptr(ell.cnt), ell.max_nbr, C.byref(term), ptr(theta), ptr(w), ptr(e),
                                         ptr(g), ptr(gth), ptr(hw), ptr(gthw), ptr(partial), float(scale),
                                         int(acc) | (2 if recheck else 0), stream_ptr(dev)), "mdg_pair_eval_ell_into")
    else:
        check(lib.mdg_pair_eval_ell(ptr(xyz), N, C.byref(ell.cell_struct), ptr(ell.col), ptr(ell.shift),
                                    ptr(ell.cnt), ell.max_nbr, C.byref(term), ptr(theta), ptr(w), ptr(e),
                                    ptr(g), ptr(gth), ptr(hw), ptr(gthw), ptr(partial), stream_ptr(dev)),
              "mdg_pair_eval_ell")
    out.update(energy=e, grad=g, gtheta=gth, hw=hw, gtheta_w=gthw)
    return out


class PairGradFn(torch.autograd.Function):
    """(dU/dx, dU/dtheta) as a differentiable op; backward = Hessian-vector product and the
    mixed theta-derivative (the second autograd pass of torchmd/sovlers.py:229-233).
    Second derivatives w.r.t. theta alone (a cotangent on dU/dtheta) are not supported."""

    @staticmethod
    def forward(ctx, xyz, theta, ell, term, cache):
        ctx.ell, ctx.term = ell, term
        ctx.save_for_backward(xyz, theta)
        ctx.set_materialize_grads(False)        # an unused dU/dtheta output arrives as None, not as zeros: no device read
        if cache is not None:
            g, gth = cache
        else:
            o = pair_eval(ell, xyz, term, theta, energy=False, grad=True)
            g, gth = o["grad"], o["gtheta"]
        if gth is None:
            gth = xyz.new_zeros(0)
        return g, gth

    @staticmethod
    def backward(ctx, wg, wgth):
        xyz, theta = ctx.saved_tensors
        if wgth is not None and wgth.numel():
            raise NotImplementedError("mdgrad_amd: second derivatives of a pair energy w.r.t. its parameters alone "
                                      "(a cotangent on dU/dtheta) are not provided by the HIP pair kernels")
        if wg is None:
            return None, None, None, None, None
        w = wg.detach().contiguous()
        o = pair_eval(ctx.ell, xyz, ctx.term, theta, w=w, energy=False, grad=False)
        gthw = o["gtheta_w"] if o["gtheta_w"] is not None else None
        return o["hw"], gthw, None, None, None


class PairEnergyFn(torch.autograd.Function):
    """U(x, theta) = sum_pairs phi(r)  (torchmd/interface.py:298-300), differentiable twice."""

    @staticmethod
    def forward(ctx, xyz, theta, ell, term):
        o = pair_eval(ell, xyz, term, theta, energy=True, grad=True)
        ctx.ell, ctx.term = ell, term
        ctx.cache = (o["grad"], o["gtheta"])
        ctx.save_for_backward(xyz, theta)
        return o["energy"].reshape(())

    @staticmethod
    def backward(ctx, gU):
        xyz, theta = ctx.saved_tensors
        g, gth = PairGradFn.apply(xyz, theta, ctx.ell, ctx.term, ctx.cache)
        gtheta = gU * gth if theta.numel() else None
        return gU * g, gtheta, None, None


# ----------------------------------------------------------------------------- bonded terms (f4)
class BondedTable:
    """Static topology table of a bonded term for mdg_bonded_eval (csrc/bonded.hip): `top` [n_terms, 2 | 3] atom indices
    (torchmd/interface.py:406-510) as int32 on the device, plus the incidence list of every atom -- entries
    4 * term + role, ascending per atom -- built once on the host."""

    def __init__(self, kind, top, n_atoms, cell_len, k, x0, device):
        import numpy as np
        width = 2 if kind == _lib.BONDED_BOND else 3
        t = torch.as_tensor(top).detach().cpu().numpy().astype(np.int64).reshape(-1, width)
        t = np.where(t < 0, t + n_atoms, t)                      # (the reference's xyz[top] accepts negative indices)
        if t.size and (t.min() < 0 or t.max() >= n_atoms):
            raise ValueError("mdgrad_amd: bonded topology refers to atoms outside [0, %d)" % n_atoms)
        self.kind, self.n_atoms, self.n_terms = int(kind), int(n_atoms), int(t.shape[0])
        atoms = t.reshape(-1)
        codes = (4 * np.repeat(np.arange(self.n_terms), width) + np.tile(np.arange(width), self.n_terms)).astype(np.int64)
        order = np.lexsort((codes, atoms))                       # by atom, then by (term, role)
        ptr_ = np.zeros(n_atoms + 1, dtype=np.int64)
        np.add.at(ptr_, atoms + 1, 1)
        i32 = dict(dtype=torch.int32, device=device)
        self.top = torch.as_tensor(t.astype(np.int32).copy(), **i32).contiguous()
        self.inc_ptr = torch.as_tensor(np.cumsum(ptr_).astype(np.int32), **i32)
        self.inc = torch.as_tensor(codes[order].astype(np.int32), **i32)
        self.cell_len = (C.c_float * 3)(*[float(x) for x in cell_len])
        self.k, self.x0 = float(k), float(x0)


def bonded_eval(tab, xyz, w=None, energy=False, grad=True, into=None, scale=1.0):
    """One launch of mdg_bonded_eval -> dict(e_atom, grad, hw).  `into` = (grad buffer, hw buffer or None): the per-atom
    outputs are ADDED onto them, times `scale` (a Stack member's force without extra launches)."""
    lib = _lib.load()
    require_gpu(xyz, "xyz")
    if xyz.shape != (tab.n_atoms, 3):
        raise ValueError("mdgrad_amd: xyz must be [%d, 3] (got %s)" % (tab.n_atoms, tuple(xyz.shape)))
    xyz = xyz.contiguous()
    dev, N = xyz.device, tab.n_atoms
    acc = into is not None
    e = torch.empty(N, device=dev) if energy else None
    g = (into[0] if acc else torch.empty(N, 3, device=dev)) if grad else None
    hw = None
    if w is not None:
        require_gpu(w, "w")
        w = w.contiguous()
        hw = into[1] if acc else torch.empty(N, 3, device=dev)
    check(lib.mdg_bonded_eval(ptr(xyz), N, tab.cell_len, tab.kind, ptr(tab.top), tab.n_terms, tab.k, tab.x0, ptr(tab.inc_ptr),
                              ptr(tab.inc), ptr(w), ptr(e), ptr(g), ptr(hw), float(scale), int(acc), stream_ptr(dev)),
          "mdg_bonded_eval")
    return dict(e_atom=e, grad=g, hw=hw)


class BondedGradFn(torch.autograd.Function):
    """dU/dx of a bonded term as a differentiable op; backward = the Hessian-vector product (the second autograd pass of
    torchmd/sovlers.py:229-233)."""

    @staticmethod
    def forward(ctx, xyz, tab, cache):
        ctx.tab = tab
        ctx.save_for_backward(xyz)
        return cache if cache is not None else bonded_eval(tab, xyz)["grad"]

    @staticmethod
    def backward(ctx, wg):
        (xyz,) = ctx.saved_tensors
        return bonded_eval(ctx.tab, xyz, w=wg.detach().contiguous(), grad=False)["hw"], None, None


class BondedEnergyFn(torch.autograd.Function):
    """U(x) of a bonded term (torchmd/interface.py:447-455, 496-508), differentiable twice."""

    @staticmethod
    def forward(ctx, xyz, tab):
        o = bonded_eval(tab, xyz, energy=True, grad=True)
        ctx.tab, ctx.cache = tab, o["grad"]
        ctx.save_for_backward(xyz)
        return o["e_atom"].sum()

    @staticmethod
    def backward(ctx, gU):
        (xyz,) = ctx.saved_tensors
        return gU * BondedGradFn.apply(xyz, ctx.tab, ctx.cache), None


# ----------------------------------------------------------------------------- fused trajectories
class FusedSpec:
    """Host-side descriptor of a fusable integrator (NoseHooverChain / NVE over built-in pair
    terms, topology_update_freq == 1)."""

    def __init__(self, ensemble, n_atoms, mass, cell_struct, terms, n_theta_total, masks,
                 T=0.0, n_dof=0.0, Q=(), block=0, large=False):
        self.large = large
        self.ensemble, self.n_atoms, self.mass = ensemble, n_atoms, mass
        self.cell_struct, self.terms, self.n_theta_total, self.masks = cell_struct, terms, n_theta_total, masks
        self.T, self.n_dof, self.Q, self.block = float(T), float(n_dof), list(Q), block

    def params(self, n_rep, n_frames):
        p = MdgTrajParams()
        p.n_rep, p.n_atoms, p.n_frames = n_rep, self.n_atoms, n_frames
        p.n_chains, p.ensemble, p.block = len(self.Q), self.ensemble, self.block
        p.T, p.n_dof = self.T, self.n_dof
        for k, q in enumerate(self.Q):
            p.Q[k] = q
        return p


def make_terms(term_list, n_theta_total):
    ts = MdgTerms()
    ts.n_terms, ts.n_theta_total = len(term_list), n_theta_total
    for k, t in enumerate(term_list):
        ts.t[k] = t
    return ts


# diagnostics of the multi-launch (large-N) trajectory path: how often the adjoint could not use the stored candidate
# lists of the forward pass (tests read these)
LARGE_STATS = {"lists_incomplete": 0, "adjoint_redone_with_searches": 0}
# the RDF of large systems searches and counts in one sweep over the cell bins (False: through a neighbour list)
RDF_CELL_DIRECT = True


def large_list_builds(spec):
    """After a forward pass of the multi-launch (large-N) kernels on `spec` (and before its workspace is released by the
    backward pass): int32 [R, T], entry (r, f) = the frame whose neighbour search produced the candidate list that
    served frame f (Verlet reuse, csrc/traj_large.hip); None when the lists were not kept.  Diagnostics:
    `len(set(row))` searches ran for that replica."""
    last = getattr(spec, "_last_large", None)
    if last is None:
        raise RuntimeError("mdgrad_amd: no trajectory of the multi-launch (large-N) kernels has run on this spec")
    ref, (R, N, T, KT) = last
    ws = ref()
    if ws is None:
        raise RuntimeError("mdgrad_amd: the trajectory's workspace has been released (ask before the backward pass ends)")
    out = torch.empty(R, T, dtype=torch.int32, device=ws.device)
    rc = _lib.load().mdg_traj_large_list_builds(ptr(ws), R, N, T, KT, ptr(out), stream_ptr(ws.device))
    if rc == 1:
        return None
    check(rc, "mdg_traj_large_list_builds")
    return out


class RdfFuse:
    """An `rdf` observable (observable.py) that a fused trajectory launch evaluates on the fly: centres, width and
    pair cutoff of the observable and the frames frame_start + k frame_stride it is called on.  Registered on the
    integrator by the observable itself the first time it sees (a time slice of) a fused trajectory; later launches
    deposit the histogram inside the trajectory kernels (mdg_traj_fwd_small_rdf) and take dL/d(raw) in the adjoint
    (mdg_traj_adj_small_rdf) -- see `fused_traj` and `rdf.forward`."""

    def __init__(self, obs, start, stride):
        import weakref
        self.obs = weakref.ref(obs)
        self.mu = obs.offsets.detach().to(torch.float32).contiguous()
        self.nbins, self.coeff, self.spacing = int(obs.nbins), float(obs.coeff), float(obs.spacing)
        self.mu0 = float(obs.r_axis[0])
        self.cutoff, self.n_atoms = float(obs.cutoff_boundary), int(obs.natoms)
        self.cell = tuple(float(x) for x in obs.cell.tolist())
        self.start, self.stride = int(start), int(stride)

    def struct(self):
        return _lib.MdgRdfFuse(self.mu.data_ptr(), self.nbins, self.coeff, self.mu0, self.spacing, self.cutoff, self.start,
                               self.stride)

    def matches(self, obs, start, stride):
        return self.obs() is obs and self.start == int(start) and self.stride == int(stride)

    def fits(self, spec, dev):
        cs = spec.cell_struct
        return (self.obs() is not None and self.n_atoms == spec.n_atoms and self.mu.device == dev
                and bool(cs.diag) and all(abs(cs.h[4 * c] - self.cell[c]) <= 1e-6 * abs(self.cell[c]) for c in range(3)))


def fused_traj(v0, q0, pv0, t, theta, spec, time_dim=None):
    """FusedTrajFn.apply plus the bookkeeping of the fused observable: returns (v_t, q_t[, pv_t]) with q_t tagged so
    that an `rdf` called on it (or on a slice of it along time) finds the histogram the launch already made, or
    registers itself for the next launch."""
    spec._fuse_now = True                  # (direct callers of FusedTrajFn.apply keep its plain output tuple)
    try:
        outs = FusedTrajFn.apply(v0, q0, pv0, t, theta, spec)
    finally:
        spec._fuse_now = False
    n = 3 if spec.ensemble == 0 else 2
    tag_trajectory(outs[1], spec, outs[n] if len(outs) > n else None,
                   (1 if v0.dim() == 3 else 0) if time_dim is None else time_dim)
    return tuple(outs[:n])


def tag_trajectory(q_t, spec, raw, time_dim):
    q_t._mdg_traj = (spec, time_dim, getattr(spec, "rdf_hint", None) if raw is not None else None, raw)


class FusedTrajFn(torch.autograd.Function):
    """odeint_adjoint(NoseHooverChain|NVE, (v0,q0[,pv0]), t) in two launches: the forward
    trajectory and, on backward, the reference's adjoint sweep (torchmd/sovlers.py:196-293).
    States may carry a leading replica dimension [R,N,3]."""

    @staticmethod
    def forward(ctx, v0, q0, pv0, t, theta, spec):
        lib = _lib.load()
        require_gpu(v0, "v0"), require_gpu(q0, "q0"), require_gpu(t, "t")
        batched = v0.dim() == 3
        R = v0.shape[0] if batched else 1
        N, T = spec.n_atoms, t.shape[0]
        dev = v0.device
        nhc = spec.ensemble == 0
        Cn = len(spec.Q)
        v0c, q0c = v0.detach().contiguous(), q0.detach().contiguous()
        pv0c = pv0.detach().contiguous() if nhc else None
        tc = t.detach().to(torch.float32).contiguous()
        thc = theta.detach().contiguous() if theta is not None and theta.numel() else None
        v_t = torch.empty(R, T, N, 3, device=dev)
        q_t = torch.empty(R, T, N, 3, device=dev)
        pv_t = torch.empty(R, T, Cn, device=dev) if nhc else None
        prm = spec.params(R, T)
        ctx.ws = None
        fuse = getattr(spec, "rdf_hint", None) if getattr(spec, "_fuse_now", False) else None
        if fuse is not None and (spec.large or getattr(spec, "table", False) or thc is None or not fuse.fits(spec, dev)
                                 or not lib.mdg_traj_rdf_supported(C.byref(prm), C.byref(spec.cell_struct),
                                                                   C.byref(spec.terms), C.byref(fuse.struct()))):
            fuse = None
        ctx.fuse = fuse
        raw = None
        if fuse is not None:
            # the observable rides along: raw soft histogram of the selected frames of all replicas
            raw = torch.empty(fuse.nbins, device=dev)
            bad = torch.zeros(R, dtype=torch.int32, device=dev)
            check(lib.mdg_traj_fwd_small_rdf(C.byref(prm), C.byref(spec.cell_struct), C.byref(spec.terms), ptr(thc),
                                             ptr(spec.mass), ptr(tc), ptr(v0c), ptr(q0c), ptr(pv0c), ptr(v_t),
                                             ptr(q_t), ptr(pv_t), ptr(bad), C.byref(fuse.struct()), ptr(raw),
                                             stream_ptr(dev)), "mdg_traj_fwd_small_rdf")
        elif spec.large:
            ws = torch.empty(int(lib.mdg_traj_large_workspace(R, N, T, spec.n_theta_total)), device=dev)
            flags = torch.zeros(8, dtype=torch.int32, device=dev)
            if getattr(spec, "stale_freq", 0):
                # topology_update_freq > 1 beyond one workgroup per replica (round 6): stale rows, the host loop of the library
                # decides per call whether it rebuilds (md.py:200-204); 2 (T - 1) calls
                integ = spec._integrator
                assert int(lib.mdg_traj_large_stale_words(R, N)) == R * N * 257
                rows = integ.stale_lists(R, N, dev, large=True)
                check(lib.mdg_traj_fwd_large_stale(C.byref(prm), C.byref(spec.cell_struct), C.byref(spec.terms), ptr(thc),
                                                   ptr(spec.mass), ptr(tc), ptr(v0c), ptr(q0c), ptr(pv0c), ptr(v_t), ptr(q_t),
                                                   ptr(pv_t), ptr(ws), ptr(flags), int(spec.stale_freq), int(integ.update_count),
                                                   ptr(rows), stream_ptr(dev)), "mdg_traj_fwd_large_stale")
                integ.update_count += 2 * (T - 1)
                integ._stale_fused_dirty = True
                ctx.stale_code = rows
            else:
                check(lib.mdg_traj_fwd_large(C.byref(prm), C.byref(spec.cell_struct), C.byref(spec.terms), ptr(thc),
                                             ptr(spec.mass), ptr(tc), ptr(v0c), ptr(q0c), ptr(pv0c), ptr(v_t),
                                             ptr(q_t), ptr(pv_t), ptr(ws), ptr(flags), stream_ptr(dev)),
                      "mdg_traj_fwd_large")
            fl = flags.tolist()                    # one sync per trajectory (neighbour buffer / table range check)
            if fl[0]:
                raise RuntimeError("mdgrad_amd: an atom has %d neighbours within the cutoff; the fused large-N "
                                   "kernel holds 256 per atom" % fl[0])
            if fl[3]:
                raise RuntimeError("mdgrad_amd: a pair came closer than the first node of the tabulated pair "
                                   "potential; lower `table_rmin` on the integrator or set `fused_table = False`")
            ctx.ws, bad = ws, flags[1:2]
            spec._last_large = (weakref.ref(ws), (R, N, T, spec.n_theta_total))    # (ops.large_list_builds)
            ctx.lists_ok = not fl[4]               # every frame's candidate list was stored: the adjoint re-tests them
            LARGE_STATS["lists_incomplete"] += int(bool(fl[4]))
        elif getattr(spec, "stale_freq", 0):
            # topology_update_freq > 1: stale lists, rebuilt at the calls whose running count is a multiple of the frequency
            # (md.py:200-204); the forward pass makes 2 (T - 1) calls
            integ = spec._integrator
            code = integ.stale_lists(R, N, dev)
            bad = torch.zeros(R, dtype=torch.int32, device=dev)
            check(lib.mdg_traj_fwd_small_stale(C.byref(prm), C.byref(spec.cell_struct), C.byref(spec.terms), ptr(thc),
                                               ptr(spec.mass), ptr(tc), ptr(v0c), ptr(q0c), ptr(pv0c), ptr(v_t), ptr(q_t),
                                               ptr(pv_t), ptr(bad), int(spec.stale_freq), int(integ.update_count), ptr(code),
                                               stream_ptr(dev)), "mdg_traj_fwd_small_stale")
            integ.update_count += 2 * (T - 1)
            integ._stale_fused_dirty = True        # (the generic path's own lists are older than these now: md.update_topology)
            ctx.stale_code = code                  # (the lists this pass ended with: what its backward continues from)
        else:
            bad = torch.zeros(R, dtype=torch.int32, device=dev)
            check(lib.mdg_traj_fwd_small(C.byref(prm), C.byref(spec.cell_struct), C.byref(spec.terms), ptr(thc),
                                         ptr(spec.mass), ptr(tc), ptr(v0c), ptr(q0c), ptr(pv0c), ptr(v_t),
                                         ptr(q_t), ptr(pv_t), ptr(bad), stream_ptr(dev)), "mdg_traj_fwd_small")
        if getattr(spec, "table", False) and not spec.large:
            # tabulated user pair module: a live pair closer than the first table node has no valid entry
            # (one host sync per trajectory, table mode only)
            if bool((bad & 2).any()):
                raise RuntimeError(
                    "mdgrad_amd: a pair came closer than table_rmin * cutoff = %.4g, below the first node of the "
                    "tabulated pair potential; lower `table_rmin` on the integrator or set `fused_table = False` "
                    "to evaluate the module per pair" % math.sqrt(spec.u0))
        ctx.spec, ctx.batched, ctx.nhc = spec, batched, nhc
        ctx.nonfinite = bad
        saved = [tc, v_t, q_t] + ([pv_t] if nhc else []) + ([thc] if thc is not None else [])
        ctx.has_theta = thc is not None
        ctx.save_for_backward(*saved)
        outs = (v_t, q_t, pv_t) if nhc else (v_t, q_t)
        if not batched:
            outs = tuple(o[0] for o in outs)
        if raw is not None:
            outs = outs + (raw,)
        return outs

    @staticmethod
    def backward(ctx, *grads):
        lib = _lib.load()
        spec, nhc = ctx.spec, ctx.nhc
        saved = list(ctx.saved_tensors)
        tc, v_t, q_t = saved[0], saved[1], saved[2]
        pv_t = saved[3] if nhc else None
        thc = saved[-1] if ctx.has_theta else None
        R, T, N = v_t.shape[0], v_t.shape[1], spec.n_atoms
        dev = v_t.device
        if ctx.nonfinite is not None and bool((ctx.nonfinite & 1).any()):
            raise RuntimeError("mdgrad_amd: the forward trajectory reached a non-finite state (time step too large "
                               "or overlapping atoms); its adjoint is undefined")

        def prep(g, like):
            if g is None:
                return None
            g = g.detach()
            if not ctx.batched:
                g = g[None]
            return g.expand_as(like).contiguous()

        gv = prep(grads[0], v_t)
        gq = prep(grads[1], q_t)
        gp = prep(grads[2], pv_t) if nhc else None
        fuse = ctx.fuse
        g_raw = grads[3 if nhc else 2] if fuse is not None else None
        if g_raw is not None and ctx.needs_input_grad[3]:
            # gradients w.r.t. the time grid read the frame gradients: materialise the observable's share
            sel = q_t[:, fuse.start::fuse.stride].contiguous()
            gsel = torch.empty_like(sel)
            cs = spec.cell_struct
            check(lib.mdg_rdf_bwd_uniform(ptr(sel), sel.shape[0] * sel.shape[1], N, C.byref(cs), fuse.cutoff, None,
                                          ptr(fuse.mu), fuse.spacing, fuse.coeff, fuse.nbins,
                                          ptr(g_raw.detach().to(torch.float32).contiguous()), ptr(gsel), stream_ptr(dev)),
                  "mdg_rdf_bwd")
            gq = torch.zeros_like(q_t) if gq is None else gq.clone()
            gq[:, fuse.start::fuse.stride] += gsel
            g_raw = None
        adj_v = torch.empty(R, N, 3, device=dev)
        adj_q = torch.empty(R, N, 3, device=dev)
        adj_p = torch.empty(R, len(spec.Q), device=dev) if nhc else None
        KT = spec.n_theta_total
        adj_th = torch.zeros(R, KT, device=dev) if KT else None
        prm = spec.params(R, T)
        table = getattr(spec, "table", False)

        ws, lists_ok = getattr(ctx, "ws", None), getattr(ctx, "lists_ok", False)

        def launch(terms, search=False):
            if g_raw is not None:
                gr = g_raw.detach().to(torch.float32).contiguous()
                check(lib.mdg_traj_adj_small_rdf(C.byref(prm), C.byref(spec.cell_struct), C.byref(terms), ptr(thc),
                                                 ptr(spec.mass), ptr(tc), ptr(v_t), ptr(q_t), ptr(pv_t), ptr(gv),
                                                 ptr(gq), ptr(gp), ptr(adj_v), ptr(adj_q), ptr(adj_p), ptr(adj_th),
                                                 C.byref(fuse.struct()), ptr(gr), stream_ptr(dev)),
                      "mdg_traj_adj_small_rdf")
                return None
            if spec.large and getattr(spec, "stale_freq", 0):
                # stale rows; 3 calls per interval (the dL/dt evaluation and two augmented ones, sovlers.py:258-266)
                integ = spec._integrator
                if getattr(integ, "_stale_code", None) is None and getattr(ctx, "stale_code", None) is not None:
                    integ._stale_code = ctx.stale_code      # (a generic call in between dropped them: see the small path below)
                rows = integ.stale_lists(R, N, dev, large=True)
                flags = torch.zeros(8, dtype=torch.int32, device=dev)
                check(lib.mdg_traj_adj_large_stale(C.byref(prm), C.byref(spec.cell_struct), C.byref(terms), ptr(thc),
                                                   ptr(spec.mass), ptr(tc), ptr(v_t), ptr(q_t), ptr(pv_t), ptr(gv), ptr(gq),
                                                   ptr(gp), ptr(adj_v), ptr(adj_q), ptr(adj_p), ptr(adj_th), ptr(ws), ptr(flags),
                                                   int(spec.stale_freq), int(integ.update_count), ptr(rows), stream_ptr(dev)),
                      "mdg_traj_adj_large_stale")
                integ.update_count += 3 * (T - 1)
                integ._stale_fused_dirty = True
                return flags
            if spec.large:
                # the stored candidate lists of the forward pass serve the adjoint unless one overflowed there (or, flag
                # 5 below, a midpoint moved past their skin): block = -1 asks for fresh searches
                # (a loop, not a recursive call: a closure that names itself is a reference cycle, and this one would
                #  keep ctx -- with the trajectory's multi-GB workspace -- alive until the cyclic collector runs)
                pl = type(prm).from_buffer_copy(prm)
                for search in ((search,) if (search or not lists_ok or spec.block == -1) else (False, True)):
                    pl.block = -1 if (search or not lists_ok or spec.block == -1) else 0
                    flags = torch.zeros(8, dtype=torch.int32, device=dev)
                    check(lib.mdg_traj_adj_large(C.byref(pl), C.byref(spec.cell_struct), C.byref(terms), ptr(thc),
                                                 ptr(spec.mass), ptr(tc), ptr(v_t), ptr(q_t), ptr(pv_t), ptr(gv), ptr(gq),
                                                 ptr(gp), ptr(adj_v), ptr(adj_q), ptr(adj_p), ptr(adj_th), ptr(ws),
                                                 ptr(flags), stream_ptr(dev)), "mdg_traj_adj_large")
                    if pl.block == -1 or not int(flags[5]):
                        break
                    LARGE_STATS["adjoint_redone_with_searches"] += 1
                return flags
            if getattr(spec, "stale_freq", 0):
                # the adjoint makes 3 calls per interval (the dL/dt evaluation and two augmented ones, sovlers.py:258-266)
                integ = spec._integrator
                if getattr(integ, "_stale_code", None) is None and getattr(ctx, "stale_code", None) is not None:
                    # a generic force / odeint call on this integrator between the fused forward and this backward (logging,
                    # an observable) dropped the fused lists (md.update_topology): the forward's own lists -- held on ctx --
                    # are the ones its saved frames were integrated with; the call counter keeps what the calls in between
                    # made of it, as the reference's would
                    integ._stale_code = ctx.stale_code
                code = integ.stale_lists(R, N, dev)
                check(lib.mdg_traj_adj_small_stale(C.byref(prm), C.byref(spec.cell_struct), C.byref(terms), ptr(thc),
                                                   ptr(spec.mass), ptr(tc), ptr(v_t), ptr(q_t), ptr(pv_t), ptr(gv), ptr(gq),
                                                   ptr(gp), ptr(adj_v), ptr(adj_q), ptr(adj_p), ptr(adj_th),
                                                   int(spec.stale_freq), int(integ.update_count), ptr(code),
                                                   stream_ptr(dev)), "mdg_traj_adj_small_stale")
                integ.update_count += 3 * (T - 1)
                integ._stale_fused_dirty = True
                return None
            check(lib.mdg_traj_adj_small(C.byref(prm), C.byref(spec.cell_struct), C.byref(terms), ptr(thc),
                                         ptr(spec.mass), ptr(tc), ptr(v_t), ptr(q_t), ptr(pv_t), ptr(gv), ptr(gq),
                                         ptr(gp), ptr(adj_v), ptr(adj_q), ptr(adj_p), ptr(adj_th),
                                         stream_ptr(dev)), "mdg_traj_adj_small")
            return None

        def check_large(flags):
            # the adjoint evaluates forces / Hessian-vector products at its own (midpoint) positions: a neighbour
            # buffer overflow or a blow-up there must not pass silently (one host sync, large path only)
            f = flags.tolist()
            if f[0]:
                raise RuntimeError("mdgrad_amd: an atom has %d neighbours within the cutoff in the adjoint sweep; "
                                   "the fused large-N kernel holds 256 per atom" % f[0])
            if f[1]:
                raise RuntimeError("mdgrad_amd: non-finite state in the fused large-N adjoint sweep")
            return f

        if table:
            # fixed-point scale of the in-kernel table-gradient scatter: the largest single contribution
            # 1/2 h (D . w_ij) is put near 2^30; the kernels accumulate int64 words and flag a single contribution at or
            # beyond 2^62 / (the contributions one word can receive) -- ~2^40 at 50 frames of 108 atoms --, so a sum cannot
            # leave the word unflagged however far the adjoint grows along the trajectory; one host sync
            lam = max([float(g.abs().max()) for g in (gv, gq) if g is not None] + [1e-30])
            h = float((tc[1:] - tc[:-1]).abs().max())
            est = max(0.5 * h * 2.0 * spec.terms.t[0].cutoff * lam / float(spec.mass.min()), 1e-30)
            S = max(-100, min(100, int(math.floor(30 - math.log2(est)))))
            for attempt in range(4):
                terms = type(spec.terms).from_buffer_copy(spec.terms)
                terms.t[0].c = 2.0 ** S
                flags = launch(terms)
                bad = bool(check_large(flags)[2]) if flags is not None else not bool(torch.isfinite(adj_th[:, 0]).all())
                if not bad:
                    break
                S -= 14                                    # adjoint grew past the range: coarser fixed point
            else:
                raise RuntimeError("mdgrad_amd: the table-gradient accumulation overflowed (adjoint magnitudes "
                                   "above ~2^%d of the incoming gradients)" % (40 - S))
        else:
            flags = launch(spec.terms)
            if flags is not None:
                check_large(flags)
        if not ctx.batched:
            adj_v, adj_q = adj_v[0], adj_q[0]
            adj_p = adj_p[0] if nhc else None
        gth = adj_th.sum(0) if adj_th is not None else None
        g_t = None
        if ctx.needs_input_grad[3]:
            g_t = _time_vjps(spec, tc, v_t, q_t, pv_t, gv, gq, gp)
        return adj_v, adj_q, adj_p, g_t, gth, None


def _time_vjps(spec, t, v_t, q_t, pv_t, gv, gq, gp):
    """dL/dt of the adjoint (`time_vjps`, torchmd/sovlers.py:258-266,289-293): dL/dt_k = f(y_k) . dL/dy_k for
    k >= 1 and dL/dt_0 = -(their sum); the right-hand side does not depend on t, so the adjoint of time is not
    changed by the augmented integration.  Nobody on the hot path consumes it (the reference's drivers do not
    either), so it is computed on request only -- one force evaluation per saved frame on the generic HIP ops."""
    integ = getattr(spec, "_integrator", None)
    if integ is None:
        raise NotImplementedError("mdgrad_amd: dL/dt needs the integrator behind the fused spec")
    R, T = v_t.shape[0], v_t.shape[1]
    nhc = pv_t is not None
    stacked = getattr(spec, "n_rep", 1) > 1
    zero = lambda like: torch.zeros_like(like)
    out = torch.zeros(T, device=v_t.device, dtype=t.dtype)
    with torch.no_grad():
        for k in range(1, T):
            frames = [None] if stacked else range(R)
            for r in frames:
                if stacked:                                     # System.replicate: the model takes the stacked state
                    v, q = v_t[:, k].reshape(-1, 3), q_t[:, k].reshape(-1, 3)
                    pv = pv_t[:, k] if nhc else None
                    g_v = gv[:, k].reshape(-1, 3) if gv is not None else zero(v)
                    g_q = gq[:, k].reshape(-1, 3) if gq is not None else zero(q)
                    g_p = (gp[:, k] if gp is not None else zero(pv)) if nhc else None
                else:
                    v, q = v_t[r, k], q_t[r, k]
                    pv = pv_t[r, k] if nhc else None
                    g_v = gv[r, k] if gv is not None else zero(v)
                    g_q = gq[r, k] if gq is not None else zero(q)
                    g_p = (gp[r, k] if gp is not None else zero(pv)) if nhc else None
                integ.model._reset_topology(q)
                F = integ.model.force(q)
                if nhc:
                    a, _, b = integ.rhs_from_force((v, q, pv), F)
                    out[k] += (a * g_v).sum() + (v * g_q).sum() + (b * g_p).sum()
                else:
                    out[k] += (F * g_v).sum() + (v * g_q).sum()              # NVE: dv/dt = F (md.py:145-148)
        out[0] = -out[1:].sum()
    return out


# ----------------------------------------------------------------------------- rdf
RDF_LIST_ATOMS = 2048        # from this size on the RDF runs over a cell-list neighbour list instead of all N^2 pairs


def _rdf_pair_table(g_raw, mu, coeff, cutoff, nodes=4096):
    """The RDF loss as a pair 'potential' phi(d) = sum_k g_k exp(coeff (d - mu_k)^2): its table for the
    MDG_PAIR_TABLE kind -- c1(u) = phi'(r)/r and du * dc1/du on a uniform grid in u = r^2 (csrc/common.hpp)."""
    lo = max(float(mu[0]) - 6.0 / math.sqrt(-2.0 * coeff), 0.05)
    u0, u1 = lo * lo, float(cutoff) ** 2
    du = (u1 - u0) / (nodes - 1)
    r = (u0 + du * torch.arange(nodes, device=mu.device, dtype=torch.float32)).sqrt()[:, None]
    x = r - mu[None, :]
    e = torch.exp(coeff * x * x) * g_raw[None, :]
    p1 = (2.0 * coeff * x * e).sum(1)                                    # phi'
    p2 = ((2.0 * coeff + (2.0 * coeff * x) ** 2) * e).sum(1)             # phi''
    r = r[:, 0]
    c1 = p1 / r
    dc1 = (p2 / r - p1 / (r * r)) / (2.0 * r)                            # d(phi'/r)/du
    return torch.stack((c1, du * dc1), 1).reshape(-1).contiguous(), u0, du


class RdfRawFn(torch.autograd.Function):
    """raw[k] = sum over frames and i<j pairs (d < cutoff) of exp(coeff (d - mu_k)^2)
    (the GaussianSmearing(...).sum(0) of torchmd/observable.py:70)."""

    @staticmethod
    def forward(ctx, xyz, mu, coeff, cutoff, cell_struct, mask, spacing=0.0, mu_last=None):
        lib = _lib.load()
        require_gpu(xyz, "xyz")
        x = xyz.detach().contiguous()
        x3 = x.reshape(-1, x.shape[-2], 3)
        F, N, B = x3.shape[0], x3.shape[1], mu.shape[0]
        dev = x.device
        raw = torch.empty(B, device=dev)
        ctx.ell = None
        list_cut = float(cutoff)
        if mu_last is not None and coeff < 0:
            # The pair search is trimmed to the reach of the Gaussians, and the bound follows the user's `width`: the
            # fine grid of the list kernels ends 5.3 / s beyond the last centre (s = sqrt(-coeff log2 e)), where a pair's
            # term in the LAST bin is exp2(-5.3^2) = 2^-28.1 of a peak term -- small, not zero: with ~ as many pairs in
            # the dropped shell as around the last centre, the last bin loses < 2^-24 of its count (half an ulp of the
            # fp32 sum) and every other bin less.  Never beyond the reference's own cutoff_boundary = end + 0.5
            # (observable.py:52), which wide Gaussians reach first (tests: width x3, x8 at 2 744 atoms vs the oracle).
            list_cut = min(list_cut, float(mu_last) + 1.02 * 5.3 / math.sqrt(-coeff * 1.4426950408889634) + 1e-6)
        if (N >= RDF_LIST_ATOMS and mask is None and spacing > 0 and cell_struct.diag
                and _use_cell_list(N, cell_struct, list_cut)
                and lib.mdg_rdf_ell_supported(float(spacing), float(coeff), B)):
            # large systems: pair search through the cell list (frames = groups of one list), every listed pair
            # counted on the fine integer grid; the gradient is a tabulated pair force over the same list
            flat = x3.reshape(F * N, 3)
            cutoff = list_cut
            muc = mu.detach().to(torch.float32).contiguous()
            if RDF_CELL_DIRECT and lib.mdg_rdf_cell_supported(N, C.byref(cell_struct), float(cutoff)):
                # ... without materialising the list: search + count in one sweep over the cell bins (csrc/rdf_cell.hip);
                # the backward pass sweeps the same bins with the tabulated pair force
                scratch = torch.empty(int(lib.mdg_rdf_cell_scratch(F, N, C.byref(cell_struct), float(cutoff))) + 4,
                                      dtype=torch.int32, device=dev)
                check(lib.mdg_rdf_fwd_cell(ptr(flat), F, N, C.byref(cell_struct), float(cutoff), ptr(muc), float(spacing),
                                           float(coeff), B, ptr(raw), ptr(scratch), stream_ptr(dev)), "mdg_rdf_fwd_cell")
                ctx.ell = scratch
                ctx.args = (float(coeff), float(cutoff), cell_struct, mask, xyz.shape, float(spacing))
                ctx.save_for_backward(flat, muc)
                return raw
            ell = build_ell(flat, cell_struct, cutoff, group=N)
            check(lib.mdg_rdf_fwd_ell(ptr(flat), F * N, C.byref(cell_struct), ptr(ell.col), ptr(ell.shift), ptr(ell.cnt),
                                      ell.max_nbr, ptr(muc), float(spacing), float(coeff), B, ptr(raw), stream_ptr(dev)),
                  "mdg_rdf_fwd_ell")
            ctx.ell = ell
            ctx.args = (float(coeff), float(cutoff), cell_struct, mask, xyz.shape, float(spacing))
            ctx.save_for_backward(flat, muc)
            return raw
        partial = torch.empty(int(lib.mdg_rdf_partial_size(F, N, B)), device=dev)
        muc = mu.detach().to(torch.float32).contiguous()
        check(lib.mdg_rdf_fwd_uniform(ptr(x3), F, N, C.byref(cell_struct), float(cutoff), ptr(mask), ptr(muc),
                                      float(spacing), float(coeff), B, ptr(raw), ptr(partial), stream_ptr(dev)),
              "mdg_rdf_fwd")
        ctx.args = (float(coeff), float(cutoff), cell_struct, mask, xyz.shape, float(spacing))
        ctx.save_for_backward(x3, muc)
        return raw

    @staticmethod
    def backward(ctx, g_raw):
        lib = _lib.load()
        x3, muc = ctx.saved_tensors
        coeff, cutoff, cell_struct, mask, shape, spacing = ctx.args
        if ctx.ell is not None:
            nodes = 4096
            table, u0, du = _rdf_pair_table(g_raw.detach().to(torch.float32), muc, coeff, cutoff, nodes)
            term = make_term(dict(kind=MDG_PAIR_TABLE, p=nodes, a=u0, phi=du, c=1.0), cutoff, 0, 2 * nodes, None)
            if torch.is_tensor(ctx.ell):                              # the forward call's cell bins (csrc/rdf_cell.hip)
                Nn = int(shape[-2])
                Fn = int(x3.shape[0]) // Nn                                # (saved positions are flat [F N, 3])
                gx = torch.empty_like(x3)
                check(lib.mdg_rdf_bwd_cell(Fn, Nn, C.byref(cell_struct), float(cutoff), C.byref(term), ptr(table),
                                           ptr(ctx.ell), ptr(gx), stream_ptr(x3.device)), "mdg_rdf_bwd_cell")
                return gx.reshape(shape), None, None, None, None, None, None, None
            o = pair_eval(ctx.ell, x3, term, table, energy=False, grad=True)
            return o["grad"].reshape(shape), None, None, None, None, None, None, None
        F, N, B = x3.shape[0], x3.shape[1], muc.shape[0]
        gx = torch.empty_like(x3)
        gr = g_raw.detach().to(torch.float32).contiguous()
        check(lib.mdg_rdf_bwd_uniform(ptr(x3), F, N, C.byref(cell_struct), cutoff, ptr(mask), ptr(muc), spacing,
                                      coeff, B, ptr(gr), ptr(gx), stream_ptr(x3.device)), "mdg_rdf_bwd")
        return gx.reshape(shape), None, None, None, None, None, None, None


# ----------------------------------------------------------------------------- velocity observables
class VacfFn(torch.autograd.Function):
    """vacf[t] = mean(v[t:] * v[:-t]) for t = 0 .. n_lags-1 (torchmd/observable.py:153-163) as one fused pass."""

    @staticmethod
    def forward(ctx, vel, n_lags):
        lib = _lib.load()
        require_gpu(vel, "vel")
        v = vel.detach().contiguous()
        T, M = v.shape[0], v[0].numel()
        out = torch.empty(n_lags, device=v.device)
        ws = torch.empty(int(lib.mdg_vacf_workspace(n_lags)), device=v.device)
        check(lib.mdg_vacf_fwd(ptr(v), T, M, n_lags, ptr(out), ptr(ws), stream_ptr(v.device)), "mdg_vacf_fwd")
        ctx.save_for_backward(v)
        ctx.n_lags = n_lags
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (v,) = ctx.saved_tensors
        gv = torch.empty_like(v)
        gc = g.detach().to(torch.float32).contiguous()
        check(lib.mdg_vacf_bwd(ptr(v), ptr(gc), v.shape[0], v[0].numel(), ctx.n_lags, ptr(gv), stream_ptr(v.device)),
              "mdg_vacf_bwd")
        return gv, None


class TemperatureFn(torch.autograd.Function):
    """Kinetic temperature sum_n m_n |v_n|^2 / n_dof of every frame of v [F, N, 3] in one launch."""

    @staticmethod
    def forward(ctx, vel, mass, n_dof):
        lib = _lib.load()
        require_gpu(vel, "vel")
        v = vel.detach().contiguous()
        F, N = v.shape[0], v.shape[1]
        out = torch.empty(F, device=v.device)
        check(lib.mdg_temperature(ptr(v), ptr(mass), F, N, float(n_dof), ptr(out), stream_ptr(v.device)), "mdg_temperature")
        ctx.save_for_backward(v, mass)
        ctx.n_dof = float(n_dof)
        return out

    @staticmethod
    def backward(ctx, g):
        v, mass = ctx.saved_tensors
        return (2.0 / ctx.n_dof) * g[:, None, None] * mass[None, :, None] * v, None, None


# ----------------------------------------------------------------------------- thermostat algebra
def nhc_rhs(v, f, pv, mass, Q, T, n_dof, n_rep, n_group):
    """(a, dpv) of NoseHooverChain.forward given the force (torchmd/md.py:221-240), one launch.
    T: device tensor [1] (read by the kernel, so that a captured graph follows update_T)."""
    lib = _lib.load()
    for t_, nm in ((v, "v"), (f, "f"), (pv, "p_v")):
        require_gpu(t_, nm)
    v, f, pv = v.contiguous(), f.contiguous(), pv.contiguous()
    a, dpv = torch.empty_like(v), torch.empty_like(pv)
    check(lib.mdg_nhc_rhs(ptr(v), ptr(f), ptr(pv), ptr(mass), ptr(Q), ptr(T), float(n_dof), int(n_rep),
                          int(n_group), int(pv.shape[-1]), ptr(a), ptr(dpv), stream_ptr(v.device)), "mdg_nhc_rhs")
    return a, dpv


def nhc_vjp(v, pv, lv, lq, lp, mass, Q, n_rep, n_group):
    """(Gv, Gp): thermostat part of the vjp of the NHC right-hand side, one launch."""
    lib = _lib.load()
    for t_, nm in ((v, "v"), (lv, "lv"), (lq, "lq")):
        require_gpu(t_, nm)
    v, pv, lv, lq, lp = v.contiguous(), pv.contiguous(), lv.contiguous(), lq.contiguous(), lp.contiguous()
    Gv, Gp = torch.empty_like(v), torch.empty_like(pv)
    check(lib.mdg_nhc_vjp(ptr(v), ptr(pv), ptr(lv), ptr(lq), ptr(lp), ptr(mass), ptr(Q), int(n_rep), int(n_group),
                          int(pv.shape[-1]), ptr(Gv), ptr(Gp), stream_ptr(v.device)), "mdg_nhc_vjp")
    return Gv, Gp


def _touched(*tensors):
    """Tensors a kernel wrote through raw pointers: bump their version counters (md._EOM.update_topology skips a
    rebuild for the same tensor object at the same version)."""
    for x in tensors:
        torch.autograd.graph.increment_version(x)


class NhvWork:
    """Buffers of the fused NH-Verlet half-step kernels (csrc/nhc.hip: mdg_nhv_*) for one integrator: the
    intermediate states of a forward step / an adjoint interval, allocated once (a captured HIP graph needs fixed
    addresses anyway)."""

    def __init__(self, integ, like_v, like_pv):
        z = lambda x: torch.empty_like(x)
        self.mass, self.Q, self.n_dof = integ.mass, integ.Q, float(integ.N_dof)
        self.R, self.n, self.C = int(integ.n_rep), int(integ.n_group), int(like_pv.shape[-1])
        self.dv_h, self.qn, self.dp_h = z(like_v), z(like_v), z(like_pv)
        self.v, self.q, self.pv, self.w = z(like_v), z(like_v), z(like_pv), z(like_v)
        self.vh, self.qm, self.pm = z(like_v), z(like_v), z(like_pv)
        self.lvh, self.lqh, self.lph, self.wh = z(like_v), z(like_v), z(like_pv), z(like_v)
        self._T = integ._T_device
        # cross-workgroup partial sums + tickets of the multi-workgroup launches (zeroed once; the kernels hand the
        # tickets back at zero)
        self.scratch = torch.zeros(int(_lib.load().mdg_nhv_scratch_floats(self.R, self.n)), device=like_v.device)

    def _a(self):
        return ptr(self.mass), ptr(self.Q), ptr(self._T()), self.n_dof

    def kick(self, v, q, pv, f, t, k):
        lib = _lib.load()
        m, Q, T, nd = self._a()
        check(lib.mdg_nhv_kick(ptr(v), ptr(q), ptr(pv), ptr(f), m, Q, T, nd, ptr(t), ptr(k), self.R, self.n, self.C,
                               ptr(self.dv_h), ptr(self.dp_h), ptr(self.qn), ptr(self.scratch), stream_ptr(v.device)),
              "mdg_nhv_kick")
        _touched(self.qn)
        return self.qn

    def finish(self, v, q, pv, f, fn, t, k, out, advance=False):
        """advance: the launch also sets k <- k + 1 (no separate increment)."""
        lib = _lib.load()
        m, Q, T, nd = self._a()
        check(lib.mdg_nhv_finish(ptr(v), ptr(q), ptr(pv), ptr(f), ptr(self.dv_h), ptr(self.dp_h), ptr(self.qn),
                                 ptr(fn.contiguous()), m, Q, T, nd, ptr(t), ptr(k), int(bool(advance)), self.R, self.n, self.C, ptr(out[0]),
                                 ptr(out[1]), ptr(out[2]), ptr(self.scratch), stream_ptr(v.device)), "mdg_nhv_finish")
        _touched(v, q, pv, f, *out)
        if advance:
            _touched(k)

    def adj_pre(self, ans, lv, i):
        lib = _lib.load()
        check(lib.mdg_nhv_adj_pre(ptr(ans[0]), ptr(ans[1]), ptr(ans[2]), ptr(lv), ptr(self.mass), ptr(i), self.R, self.n,
                                  self.C, ptr(self.v), ptr(self.q), ptr(self.pv), ptr(self.w), stream_ptr(lv.device)),
              "mdg_nhv_adj_pre")
        _touched(self.q, self.w)
        return self.q, self.w

    def adj_mid(self, lam, f, dwf, t, i):
        lib = _lib.load()
        m, Q, T, nd = self._a()
        check(lib.mdg_nhv_adj_mid(ptr(self.v), ptr(self.q), ptr(self.pv), ptr(lam[0]), ptr(lam[1]), ptr(lam[2]),
                                  ptr(f.contiguous()), ptr(dwf.contiguous()), m, Q, T, nd, ptr(t), ptr(i), self.R, self.n,
                                  self.C, ptr(self.vh), ptr(self.qm), ptr(self.pm), ptr(self.lvh), ptr(self.lqh),
                                  ptr(self.lph), ptr(self.wh), ptr(self.scratch), stream_ptr(f.device)), "mdg_nhv_adj_mid")
        _touched(self.qm, self.wh)
        return self.qm, self.wh

    def adj_end(self, lam, dwf, t, i, gout, advance=False):
        """advance: the launch also sets i <- i - 1."""
        lib = _lib.load()
        check(lib.mdg_nhv_adj_end(ptr(self.vh), ptr(self.pm), ptr(self.lvh), ptr(self.lqh), ptr(self.lph),
                                  ptr(dwf.contiguous()), ptr(self.mass), ptr(self.Q), ptr(t), ptr(i), int(bool(advance)), ptr(gout[0]),
                                  ptr(gout[1]), ptr(gout[2]), self.R, self.n, self.C, ptr(lam[0]), ptr(lam[1]), ptr(lam[2]),
                                  ptr(self.scratch), stream_ptr(dwf.device)), "mdg_nhv_adj_end")
        _touched(*lam)
        if advance:
            _touched(i)


# ----------------------------------------------------------------------------- graph ops (SchNet)
class GraphTopo:
    """Edge topology for the message-passing kernels: ELL list + undirected edge ids + the
    reference-format half list (nbr int64 [E,2], raw image flags [E,3])."""

    def __init__(self, ell):
        self.ell = ell
        self.nbr, self.offsets, self.eid = ell.half_list(with_edge_id=True)
        self.n_atoms, self.n_edges = ell.n_atoms, int(self.nbr.shape[0])
        self._bucketed = None

    def bucketed(self, bucket):
        """The same topology with the edge arrays padded (inert rows, see StaticTopo) to a multiple of
        `bucket`: the edge-wise GEMM shapes then repeat from one neighbour rebuild to the next, which is
        what lets the hipBLASLt kernels (2x faster than rocBLAS on [E,128] operands) be used without
        paying its per-shape heuristic search at every call."""
        if self._bucketed is None or self._bucketed.n_edges % bucket:
            cap = (self.n_edges + bucket - 1) // bucket * bucket
            t = object.__new__(GraphTopo)
            t.ell, t.eid, t.n_atoms, t.n_edges, t._bucketed = self.ell, self.eid, self.n_atoms, cap, None
            t.nbr = torch.full((cap, 2), -1, dtype=torch.int64, device=self.nbr.device)
            t.nbr[:self.n_edges] = self.nbr
            t.offsets = torch.zeros(cap, 3, device=self.nbr.device)
            t.offsets[:self.n_edges] = self.offsets
            t.offsets[self.n_edges:, 0] = StaticTopo.PAD_OFFSET
            t.padded = True
            self._bucketed = t
        return self._bucketed


class StaticTopo:
    """GraphTopo with a fixed edge capacity (see EllList.half_list_padded): every edge-wise tensor has
    `capacity` rows whatever the current pair count, so the whole evaluation can be captured into a
    HIP graph and replayed after each neighbour rebuild."""

    PAD_OFFSET = 1.0e4            # raw image flag of a padding row: |delta| = 1e4, every Gaussian is exactly 0

    padded = True

    def __init__(self, ell, capacity, need):
        self.ell = ell
        self.nbr, self.offsets, self.eid, self.n_valid = ell.half_list_padded(capacity, self.PAD_OFFSET, need)
        self.n_atoms, self.n_edges = ell.n_atoms, int(capacity)


class VerletList:
    """Persistent fixed-capacity neighbour list with Verlet reuse (mdg_nbr_verlet_rebuild, csrc/nbr.hip): searched with
    cutoff + skin, kept while every atom stays within skin / 2 of where it was built (decided on the device, no host
    sync), consumed through the exact per-pair cutoff test (`edge_geom` masks, `pair_eval` re-checks) -- so every
    evaluation sees the pair set a fresh search at the cutoff would find.  All buffers live as long as the object: a
    captured HIP graph replays the same addresses."""

    def __init__(self, n_atoms, group, cell_struct, cutoff, skin, mask, max_nbr, capacity, device):
        lib = _lib.load()
        self.n_atoms, self.group, self.cell_struct = int(n_atoms), int(group), cell_struct
        self.cutoff, self.skin, self.mask = float(cutoff), float(skin), mask
        self.list_cutoff = self.cutoff + self.skin
        self.max_nbr, self.capacity = int(max_nbr), int(capacity)
        self.sig = (self.max_nbr, self.capacity)
        i32 = dict(dtype=torch.int32, device=device)
        self.col, self.shift = torch.zeros(n_atoms, max_nbr, **i32), torch.zeros(n_atoms, max_nbr, **i32)
        self.cnt = torch.zeros(n_atoms, **i32)
        self.nbr = torch.full((capacity, 2), -1, dtype=torch.int64, device=device)
        self.offsets = torch.zeros(capacity, 3, device=device)
        self.offsets[:, 0] = StaticTopo.PAD_OFFSET
        self.eid = torch.zeros(n_atoms, max_nbr, **i32)
        self.n_valid = torch.zeros(1, **i32)
        self.row_base = torch.zeros(n_atoms + 1, **i32)
        self.pos_build = torch.full((n_atoms, 3), float("nan"), device=device)
        self.state = torch.zeros(4, **i32)
        self.use_cell = bool(_use_cell_list(group, cell_struct, self.list_cutoff) and max_nbr <= 512)
        ns = int(lib.mdg_nbr_cell_scratch_groups(n_atoms, group, C.byref(cell_struct), self.list_cutoff)) if self.use_cell else 1
        self.scratch = torch.zeros(max(1, ns), **i32)
        self.ell = EllList(n_atoms, max_nbr, self.col, self.shift, self.cnt, cell_struct, self.cutoff, mask)
        self.ell.verlet = True
        self.topo = _VerletTopo(self)

    def rebuild(self, xyz, need):
        """Decide on the device whether the stored list still serves `xyz`; search again if not."""
        lib = _lib.load()
        x = xyz.detach().contiguous()
        check(lib.mdg_nbr_verlet_rebuild(ptr(x), self.n_atoms, self.group, C.byref(self.cell_struct), self.list_cutoff,
                                         0.5 * self.skin, ptr(self.mask), int(self.use_cell), ptr(self.col), ptr(self.shift),
                                         ptr(self.cnt), self.max_nbr, self.capacity, float(StaticTopo.PAD_OFFSET), ptr(self.nbr),
                                         ptr(self.offsets), ptr(self.eid), ptr(self.n_valid), ptr(need), ptr(self.pos_build),
                                         ptr(self.state), ptr(self.row_base), ptr(self.scratch), stream_ptr(x.device)),
              "mdg_nbr_verlet_rebuild")

    def builds(self):
        """Searches so far (one host sync; diagnostics and tests)."""
        return int(self.state[3].item())


class _VerletTopo:
    """The StaticTopo view of a VerletList (same attributes; `verlet` tells the geometry kernel to mask)."""

    padded = True

    def __init__(self, vl):
        self.ell, self.nbr, self.offsets, self.eid, self.n_valid = vl.ell, vl.nbr, vl.offsets, vl.eid, vl.n_valid
        self.n_atoms, self.n_edges = vl.n_atoms, vl.capacity
        self.verlet = (vl.cell_struct, vl.cutoff)


def _edge_diff(x, topo):
    lib = _lib.load()
    require_gpu(x, "x")
    x = x.contiguous()
    out = torch.empty(topo.n_edges, x.shape[1], device=x.device)
    check(lib.mdg_edge_diff(ptr(x), ptr(topo.nbr), topo.n_edges, x.shape[1], ptr(out), stream_ptr(x.device)),
          "mdg_edge_diff")
    return out


def _edge_scatter(g, topo):
    lib = _lib.load()
    require_gpu(g, "g")
    g = g.contiguous()
    e = topo.ell
    out = torch.empty(topo.n_atoms, g.shape[1], device=g.device)
    check(lib.mdg_edge_scatter(ptr(g), ptr(e.col), ptr(topo.eid), ptr(e.cnt), topo.n_atoms, e.max_nbr, g.shape[1],
                               ptr(out), stream_ptr(g.device)), "mdg_edge_scatter")
    return out


def _cfconv_agg(h, W, topo):
    lib = _lib.load()
    require_gpu(h, "h"), require_gpu(W, "W")
    h, W = h.contiguous(), W.contiguous()
    e = topo.ell
    out = torch.empty(topo.n_atoms, h.shape[1], device=h.device)
    check(lib.mdg_cfconv_agg(ptr(h), ptr(W), ptr(e.col), ptr(topo.eid), ptr(e.cnt), topo.n_atoms, e.max_nbr,
                             h.shape[1], ptr(out), stream_ptr(h.device)), "mdg_cfconv_agg")
    return out


def _edge_prod(a, b, topo):
    lib = _lib.load()
    require_gpu(a, "a"), require_gpu(b, "b")
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty(topo.n_edges, a.shape[1], device=a.device)
    check(lib.mdg_edge_prod(ptr(a), ptr(b), ptr(topo.nbr), topo.n_edges, a.shape[1], ptr(out),
                            stream_ptr(a.device)), "mdg_edge_prod")
    return out


class EdgeDiffFn(torch.autograd.Function):
    """out[e] = x[i_e] - x[j_e]  (linear; transpose = EdgeScatterFn)."""

    @staticmethod
    def forward(ctx, x, topo):
        ctx.topo = topo
        return _edge_diff(x.detach(), topo)

    @staticmethod
    def backward(ctx, g):
        return EdgeScatterFn.apply(g, ctx.topo), None


class EdgeScatterFn(torch.autograd.Function):
    """out[n] = sum_{e: i_e = n} g[e] - sum_{e: j_e = n} g[e]  (transpose of EdgeDiffFn)."""

    @staticmethod
    def forward(ctx, g, topo):
        ctx.topo = topo
        return _edge_scatter(g.detach(), topo)

    @staticmethod
    def backward(ctx, gn):
        return EdgeDiffFn.apply(gn, ctx.topo), None


class CfconvAggFn(torch.autograd.Function):
    """m[n] = sum over neighbours j of n of h[j] * W[edge(n,j)]: the cfconv message plus both
    scatter_adds of nff/nn/graphconv.py:43-53 in one gather (bilinear, closed under d/d.)."""

    @staticmethod
    def forward(ctx, h, W, topo):
        ctx.topo = topo
        ctx.save_for_backward(h, W)
        return _cfconv_agg(h.detach(), W.detach(), topo)

    @staticmethod
    def backward(ctx, gm):
        h, W = ctx.saved_tensors
        gh = CfconvAggFn.apply(gm, W, ctx.topo) if ctx.needs_input_grad[0] else None
        gW = EdgeProdFn.apply(h, gm, ctx.topo) if ctx.needs_input_grad[1] else None
        return gh, gW, None


class EdgeProdFn(torch.autograd.Function):
    """out[e] = a[i_e] b[j_e] + a[j_e] b[i_e]."""

    @staticmethod
    def forward(ctx, a, b, topo):
        ctx.topo = topo
        ctx.save_for_backward(a, b)
        return _edge_prod(a.detach(), b.detach(), topo)

    @staticmethod
    def backward(ctx, gE):
        a, b = ctx.saved_tensors
        ga = CfconvAggFn.apply(b, gE, ctx.topo) if ctx.needs_input_grad[0] else None
        gb = CfconvAggFn.apply(a, gE, ctx.topo) if ctx.needs_input_grad[1] else None
        return ga, gb, None


# ----------------------------------------------------------------------------- dense algebra closure
def _atb(A, B):
    lib = _lib.load()
    require_gpu(A, "A"), require_gpu(B, "B")
    A, B = A.contiguous(), B.contiguous()
    tops = _torch_ops.get()
    if tops is not None:
        return tops.atb(A, B)
    E, M, N = A.shape[0], A.shape[1], B.shape[1]
    out = torch.empty(M, N, device=A.device)
    ws = torch.empty(max(1, int(lib.mdg_atb_workspace(E, M, N))), device=A.device)
    check(lib.mdg_atb(ptr(A), ptr(B), E, M, N, ptr(out), ptr(ws), stream_ptr(A.device)), "mdg_atb")
    return out


def _atb2(A, B, A2, B2):
    """A^T B + A2^T B2 (same shapes) in one split-K launch pair (csrc/atb.hip mdg_atb2)."""
    lib = _lib.load()
    for x, nm in ((A, "A"), (B, "B"), (A2, "A2"), (B2, "B2")):
        require_gpu(x, nm)
    A, B, A2, B2 = A.contiguous(), B.contiguous(), A2.contiguous(), B2.contiguous()
    E, M, N = A.shape[0], A.shape[1], B.shape[1]
    assert A2.shape == A.shape and B2.shape == B.shape, "atb2: the two pairs must have the same shapes"
    out = torch.empty(M, N, device=A.device)
    ws = torch.empty(max(1, int(lib.mdg_atb_workspace(E, M, N))), device=A.device)
    check(lib.mdg_atb2(ptr(A), ptr(B), ptr(A2), ptr(B2), E, M, N, ptr(out), ptr(ws), stream_ptr(A.device)), "mdg_atb2")
    return out


def _mm(A, W):
    """A[E,K] @ W[K,N] on the hand-written MFMA node kernel (mdg_dense, csrc/dense.hip) -- round 6: no library GEMM behind
    `Dense` on the autograd path either (nff/nn/layers.py:86-134) -- for 2-D f32 HIP operands with K a multiple of 4; W may be
    the transposed view of a Linear weight (read in place, Linear layout) or a contiguous [K,N] matrix.  Anything else: matmul."""
    if (A.is_cuda and A.dim() == 2 and W.dim() == 2 and A.dtype == torch.float32 and W.dtype == torch.float32
            and A.shape[0] > 0 and A.shape[1] % 4 == 0 and A.shape[1] <= DENSE_MAX_K and A.is_contiguous()):
        if W.t().is_contiguous():
            return dense(W.t(), A)[0]
        return dense(W.contiguous(), A, trans=True)[0]
    return A.matmul(W)


class MMFn(torch.autograd.Function):
    """A[E,K] @ W[K,N] on the MFMA node kernel (`_mm`); its weight gradient is the tall-skinny
    A^T g, which goes to AtBFn.  {MMFn, AtBFn} is closed under differentiation."""

    @staticmethod
    def forward(ctx, A, W):
        ctx.save_for_backward(A, W)
        return _mm(A.detach(), W.detach())

    @staticmethod
    def backward(ctx, g):
        A, W = ctx.saved_tensors
        gA = MMFn.apply(g, W.t()) if ctx.needs_input_grad[0] else None
        gW = AtBFn.apply(A, g) if ctx.needs_input_grad[1] else None
        return gA, gW


class AtBFn(torch.autograd.Function):
    """A[E,M]^T @ B[E,N] with split-K over the edges on the f32 MFMA (csrc/atb.hip)."""

    @staticmethod
    def forward(ctx, A, B):
        ctx.save_for_backward(A, B)
        if A.is_cuda and A.dtype == torch.float32:
            return _atb(A.detach(), B.detach())
        return A.detach().t().matmul(B.detach())

    @staticmethod
    def backward(ctx, g):
        A, B = ctx.saved_tensors
        gA = MMFn.apply(B, g.t()) if ctx.needs_input_grad[0] else None
        gB = MMFn.apply(A, g) if ctx.needs_input_grad[1] else None
        return gA, gB


TALL_ROWS = 8192      # (analytic.py: from here on a^T b goes to the split-K kernel)


def linear(x, weight, bias=None):
    """F.linear for 2-D f32 HIP x on the hand-written kernels: the product on the MFMA node kernel (`_mm`), the weight
    gradient x^T g on the split-K kernel (AtBFn); other inputs (CPU, other dtypes, widths that are no multiple of 4) take
    torch's own."""
    if not (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.shape[1] % 4 == 0
            and x.shape[0] > 0):
        return torch.nn.functional.linear(x, weight, bias)
    y = MMFn.apply(x, weight.t())
    return y if bias is None else y + bias


def mm(a, w):
    return MMFn.apply(a, w) if a.shape[0] >= TALL_ROWS else a.matmul(w)


def atb(a, b):
    return AtBFn.apply(a, b) if a.shape[0] >= TALL_ROWS else a.t().matmul(b)


# ----------------------------------------------------------------------------- cfconv filter (MFMA)
def filter_reference(d, mu, width, W1, b1, W2, b2):
    """The filter network in plain torch ops (nff/nn/modules.py:531-541): used for the
    derivative formulas' cross-check in tests and for inputs the fused kernel does not take."""
    g = torch.exp(-0.5 / width.pow(2) * (d[:, None] - mu).pow(2))
    h1 = torch.nn.functional.softplus(torch.nn.functional.linear(g, W1, b1)) - math.log(2.0)
    return torch.nn.functional.linear(h1, W2, b2)


class CfconvFilterFn(torch.autograd.Function):
    """W[E,F] = Dense2(ssp(Dense1(smear(d)))) in one MFMA kernel (csrc/cfconv_filter.hip).
    backward is written out with torch ops (GEMMs on rocBLAS), so it is differentiable again:
    the adjoint's second-order pass goes through it without a hand-derived third kernel."""

    @staticmethod
    def forward(ctx, d, mu, width, W1, b1, W2, b2, bf16=False):
        lib = _lib.load()
        require_gpu(d, "d")
        args = [x.detach().contiguous() for x in (d, mu, width, W1, b1, W2, b2)]
        E, G, F = args[0].shape[0], args[1].shape[0], args[5].shape[0]
        out = torch.empty(E, F, device=d.device)
        fn = lib.mdg_cfconv_filter_bf16 if bf16 else lib.mdg_cfconv_filter
        check(fn(ptr(args[0]), E, ptr(args[1]), ptr(args[2]), G, ptr(args[3]), ptr(args[4]),
                 ptr(args[5]), ptr(args[6]), F, ptr(out), stream_ptr(d.device)), "mdg_cfconv_filter")
        ctx.save_for_backward(d, mu, width, W1, b1, W2, b2)
        return out

    @staticmethod
    def backward(ctx, gW):
        d, mu, width, W1, b1, W2, b2 = ctx.saved_tensors
        need = ctx.needs_input_grad
        c = -0.5 / width.pow(2)
        x = d[:, None] - mu
        g = torch.exp(c * x.pow(2))
        a1 = linear(g, W1, b1)
        gh1 = mm(gW, W2)
        ga1 = gh1 * torch.sigmoid(a1)
        gd = gmu = gwidth = gW1 = gb1 = gW2 = gb2 = None
        if need[0] or need[1] or need[2]:
            gg = mm(ga1, W1) * g
            if need[0] or need[1]:
                t = gg * (2 * c * x)
                gd = t.sum(1) if need[0] else None
                gmu = -t.sum(0) if need[1] else None
            if need[2]:
                gwidth = (gg * x.pow(2)).sum(0) / width.pow(3)
        if need[3]:
            gW1 = atb(ga1, g)
        if need[4]:
            gb1 = ga1.sum(0)
        if need[5]:
            h1 = torch.nn.functional.softplus(a1) - math.log(2.0)
            gW2 = atb(gW, h1)
        if need[6]:
            gb2 = gW.sum(0)
        return gd, gmu, gwidth, gW1, gb1, gW2, gb2, None


# ----------------------------------------------------------------------------- fused interaction block
class FilterNet:
    """Device-side description of one SchNetConv filter network for the fused kernels (csrc/cfconv_fused.hip):
    Gaussian centres / coefficients and the two Dense layers, as contiguous fp32 tensors kept alive here."""

    bf16_reverse = True                          # ... and in the reverse sweep of the filter network (mdg_cfconv_bwd_bf16)

    def __init__(self, mu, coef, W1, b1, W2, b2, bf16=False, rows16=False):
        self.bf16 = bool(bf16)                   # bf16 MFMA operands in the forward / tangent / aggregation sweeps
        # ... and bf16 MIRRORS of the gathered node matrices (mdg_cfconv_*_rows16: a precision option of its own, see
        # include/mdgrad_hip.h); the callers hand cfconv_fwd / cfconv_bwd torch.bfloat16 matrices then
        self.rows16 = bool(rows16 and bf16 and self.bf16_reverse and
                           _lib.load().mdg_cfconv_rows16_supported(int(mu.shape[0]), int(W2.shape[0])))
        # the reverse sweep with parameter gradients can hand out d/d b2 as well (a spare padded column, mdg_cfconv_bwd_theta)
        self.b2col = bool(_lib.load().mdg_cfconv_bias_column(int(mu.shape[0])))
        self.t = [x.detach().to(torch.float32).contiguous() for x in (mu, coef, W1, b1, W2, b2)]
        self.G, self.F = int(self.t[0].shape[0]), int(self.t[4].shape[0])
        self.mu, self.coef, self.W1, self.b1, self.W2, self.b2 = self.t
        s = _lib.MdgFilterNet()
        s.mu, s.coef, s.W1, s.b1, s.W2, s.b2 = (x.data_ptr() for x in self.t)
        s.n_gauss, s.n_filters = self.G, self.F
        self.struct = s

    @staticmethod
    def supported(n_gauss, n_filters):
        return bool(_lib.load().mdg_cfconv_supported(int(n_gauss), int(n_filters)))


def edge_geom(x, topo, w=None):
    """(d, uhat[, dd, ddel]) per edge of the half list: nff/nn/models/schnet.py:142 and its tangent along w."""
    lib = _lib.load()
    require_gpu(x, "x")
    x = x.contiguous()
    verlet = getattr(topo, "verlet", None)       # a list searched with a skin: pairs beyond the cutoff get d = -1
    tops = _torch_ops.get()
    if tops is not None and verlet is None:
        d, uhat, dd, ddel = tops.edge_geom(x, w.contiguous() if w is not None else None, topo.nbr, topo.offsets)
        return (d, uhat, dd, ddel) if w is not None else (d, uhat, None, None)
    E, dev = topo.n_edges, x.device
    d, uhat = torch.empty(E, device=dev), torch.empty(E, 3, device=dev)
    dd = ddel = None
    if w is not None:
        w = w.contiguous()
        dd, ddel = torch.empty(E, device=dev), torch.empty(E, 3, device=dev)
    if verlet is not None:
        check(lib.mdg_edge_geom_masked(ptr(x), ptr(w), ptr(topo.nbr), ptr(topo.offsets), E, C.byref(verlet[0]), float(verlet[1]),
                                       ptr(d), ptr(uhat), ptr(dd), ptr(ddel), stream_ptr(dev)), "mdg_edge_geom_masked")
        return d, uhat, dd, ddel
    check(lib.mdg_edge_geom(ptr(x), ptr(w), ptr(topo.nbr), ptr(topo.offsets), E, ptr(d), ptr(uhat), ptr(dd), ptr(ddel),
                            stream_ptr(dev)), "mdg_edge_geom")
    return d, uhat, dd, ddel


def edge_geom_bwd(d_b, dd_b, d, dd, uhat, ddel, topo):
    """(F, d(w.F)/dx) per atom from the per-edge adjoints (d_b None: force only)."""
    lib = _lib.load()
    e = topo.ell
    dev = dd_b.device
    tops = _torch_ops.get()
    if tops is not None:
        force, dwf = tops.edge_geom_bwd(d_b, dd_b, d, dd, uhat, ddel, e.col, topo.eid, e.cnt)
        return force, (dwf if d_b is not None else None)
    force = torch.empty(topo.n_atoms, 3, device=dev)
    dwf = torch.empty(topo.n_atoms, 3, device=dev) if d_b is not None else None
    check(lib.mdg_edge_geom_bwd(ptr(d_b), ptr(dd_b), ptr(d), ptr(dd), ptr(uhat), ptr(ddel), ptr(e.col), ptr(topo.eid),
                                ptr(e.cnt), topo.n_atoms, e.max_nbr, ptr(force), ptr(dwf), stream_ptr(dev)),
          "mdg_edge_geom_bwd")
    return force, dwf


def cfconv_fwd(fnet, d, dd, h, hd, topo, want_sums=False):
    """(m, md, hsum, hdsum): filter generation + gather-multiply + per-atom sum in one kernel; with dd the
    forward-mode tangent rides along (hd may be None: no node tangent yet)."""
    lib = _lib.load()
    require_gpu(h, "h", dtype=None)
    if h.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("mdgrad_amd: h must be float32 (or a bfloat16 mirror for the rows16 kernels), got %s" % h.dtype)
    h = h.contiguous()
    hd = hd.contiguous() if hd is not None else None
    e = topo.ell
    N, dev = topo.n_atoms, h.device
    r16 = h.dtype == torch.bfloat16               # bf16 mirrors of the node rows: the rows16 kernels
    if r16:
        assert fnet.bf16 and (hd is None or hd.dtype == torch.bfloat16), "cfconv_fwd: bf16 node rows go with the bf16 filter kernels"
    tops = _torch_ops.get()
    if tops is not None:
        m, md, hsum, hdsum = tops.cfconv_fwd(fnet.mu, fnet.coef, fnet.W1, fnet.b1, fnet.W2, fnet.b2, bool(fnet.bf16), d, dd,
                                             h, hd, e.col, topo.eid, e.cnt, bool(want_sums))
        return (m, md if dd is not None else None, hsum if want_sums else None,
                hdsum if (want_sums and hd is not None) else None)
    m = torch.empty(N, fnet.F, device=dev)
    md = torch.empty(N, fnet.F, device=dev) if dd is not None else None
    hsum = torch.empty(N, fnet.F, device=dev) if want_sums else None
    hdsum = torch.empty(N, fnet.F, device=dev) if (want_sums and hd is not None) else None
    fn = lib.mdg_cfconv_fwd_rows16 if r16 else (lib.mdg_cfconv_fwd_bf16 if fnet.bf16 else lib.mdg_cfconv_fwd)
    check(fn(C.byref(fnet.struct), ptr(d), ptr(dd), ptr(h), ptr(hd), ptr(e.col), ptr(topo.eid), ptr(e.cnt),
             N, e.max_nbr, ptr(m), ptr(md), ptr(hsum), ptr(hdsum), stream_ptr(dev)), "mdg_cfconv_fwd")
    return m, md, hsum, hdsum


def cfconv_filter_stash(fnet, d, dd, topo):
    """(st_s, st_sd): the first Dense layer of the filter network once per undirected edge (mdg_cfconv_filter_stash) --
    [E, W] bf16 rows of s = ssp(a) and, with dd, of the tangent sd = sigmoid(a) a_dot; st_sd is None without dd."""
    lib = _lib.load()
    assert fnet.bf16, "the stash holds the bf16 operands of the bf16 kernels"
    E, dev = topo.n_edges, d.device
    W = int(lib.mdg_cfconv_stash_width(fnet.G))
    st_s = torch.empty(E, W, device=dev, dtype=torch.bfloat16)
    st_sd = torch.empty(E, W, device=dev, dtype=torch.bfloat16) if dd is not None else None
    check(lib.mdg_cfconv_filter_stash(C.byref(fnet.struct), ptr(d), ptr(dd), E, ptr(getattr(topo, "n_valid", None)), ptr(st_s),
                                      ptr(st_sd), stream_ptr(dev)), "mdg_cfconv_filter_stash")
    return st_s, st_sd


def cfconv_fwd_stashed(fnet, st_s, st_sd, d, h, hd, topo):
    """(m, md) of cfconv_fwd with the second filter layer's operands read from the stash (bitwise the same outputs); the
    tangent sweep iff st_sd is given."""
    lib = _lib.load()
    e = topo.ell
    N, dev = topo.n_atoms, h.device
    r16 = h.dtype == torch.bfloat16
    m = torch.empty(N, fnet.F, device=dev)
    md = torch.empty(N, fnet.F, device=dev) if st_sd is not None else None
    check(lib.mdg_cfconv_fwd_stashed(C.byref(fnet.struct), ptr(st_s), ptr(st_sd), ptr(d), ptr(h.contiguous()),
                                     ptr(hd.contiguous()) if hd is not None else None, ptr(e.col), ptr(topo.eid), ptr(e.cnt), N,
                                     e.max_nbr, ptr(m), ptr(md), int(r16), stream_ptr(dev)), "mdg_cfconv_fwd_stashed")
    return m, md


def cfconv_bwd(fnet, d, dd, topo, h, hd, mb, mdb, d_b, dd_b, want_theta=False, want_smear=False, want_b2=False):
    """Adjoint of the filter network (see include/mdgrad_hip.h): accumulates into d_b / dd_b in place and returns
    (gW1, gb1, gW2) when want_theta -- plus (gmu, gcoef), the gradients of the Gaussian centres and coefficients, when
    want_smear (trainable radial basis) -- plus gb2, the gradient of the second layer's bias, LAST when want_b2
    (`fnet.b2col`: mdg_cfconv_bwd_theta)."""
    lib = _lib.load()
    dev = h.device
    h, mdb = h.contiguous(), mdb.contiguous()
    hd = hd.contiguous() if hd is not None else None
    mb = mb.contiguous() if mb is not None else None
    bf16 = bool(fnet.bf16 and fnet.bf16_reverse)
    r16 = h.dtype == torch.bfloat16               # bf16 mirrors of the four gathered matrices: mdg_cfconv_bwd_rows16
    if r16:
        assert bf16 and all(t is None or t.dtype == torch.bfloat16 for t in (hd, mb, mdb)), \
            "cfconv_bwd: bf16 node rows come for all gathered matrices, with the bf16 filter kernels"
    if want_b2:
        assert want_theta and fnet.b2col, "cfconv_bwd: the bias gradient comes with the parameter gradients (FilterNet.b2col)"
        gW1, gb1 = torch.empty(fnet.G, fnet.G, device=dev), torch.empty(fnet.G, device=dev)
        gW2, gb2 = torch.empty(fnet.F, fnet.G, device=dev), torch.empty(fnet.F, device=dev)
        gmu = gcf = None
        if want_smear:
            gmu, gcf = torch.empty(fnet.G, device=dev), torch.empty(fnet.G, device=dev)
        ws = torch.empty(max(1, int(lib.mdg_cfconv_bwd_workspace(fnet.G, fnet.F, topo.n_edges))), device=dev)
        flags = (_lib.CFCONV_BF16 if bf16 else 0) | (_lib.CFCONV_ROWS16 if r16 else 0)
        check(lib.mdg_cfconv_bwd_theta(C.byref(fnet.struct), ptr(d), ptr(dd), ptr(topo.nbr), topo.n_edges, int(h.shape[0]), ptr(h),
                                       ptr(hd), ptr(mb), ptr(mdb), ptr(d_b), ptr(dd_b), ptr(gW1), ptr(gb1), ptr(gW2), ptr(gb2),
                                       ptr(gmu), ptr(gcf), ptr(ws), ptr(getattr(topo, "n_valid", None)), flags, stream_ptr(dev)),
              "mdg_cfconv_bwd_theta")
        return (gW1, gb1, gW2, gmu, gcf, gb2) if want_smear else (gW1, gb1, gW2, gb2)
    tops = _torch_ops.get()
    if tops is not None and not want_smear:
        out = tops.cfconv_bwd(fnet.mu, fnet.coef, fnet.W1, fnet.b1, fnet.W2, fnet.b2, d, dd, topo.nbr, int(topo.n_edges), h, hd,
                              mb, mdb, d_b, dd_b, getattr(topo, "n_valid", None), bool(want_theta), bf16)
        return tuple(out) if want_theta else None
    gW1 = gb1 = gW2 = gmu = gcf = ws = None
    if want_theta:
        gW1, gb1 = torch.empty(fnet.G, fnet.G, device=dev), torch.empty(fnet.G, device=dev)
        gW2 = torch.empty(fnet.F, fnet.G, device=dev)
        ws = torch.empty(max(1, int(lib.mdg_cfconv_bwd_workspace(fnet.G, fnet.F, topo.n_edges))), device=dev)
    nv = ptr(getattr(topo, "n_valid", None))
    if want_smear:
        assert want_theta, "the basis gradients come with the parameter gradients"
        gmu, gcf = torch.empty(fnet.G, device=dev), torch.empty(fnet.G, device=dev)
    if r16:
        check(lib.mdg_cfconv_bwd_rows16(C.byref(fnet.struct), ptr(d), ptr(dd), ptr(topo.nbr), topo.n_edges, int(h.shape[0]), ptr(h),
                                        ptr(hd), ptr(mb), ptr(mdb), ptr(d_b), ptr(dd_b), ptr(gW1), ptr(gb1), ptr(gW2), ptr(gmu),
                                        ptr(gcf), ptr(ws), nv, stream_ptr(dev)), "mdg_cfconv_bwd_rows16")
        return ((gW1, gb1, gW2, gmu, gcf) if want_smear else (gW1, gb1, gW2)) if want_theta else None
    if want_smear:
        check(lib.mdg_cfconv_bwd_smear(C.byref(fnet.struct), ptr(d), ptr(dd), ptr(topo.nbr), topo.n_edges, ptr(h), ptr(hd), ptr(mb),
                                       ptr(mdb), ptr(d_b), ptr(dd_b), ptr(gW1), ptr(gb1), ptr(gW2), ptr(gmu), ptr(gcf), ptr(ws), nv,
                                       int(bf16), stream_ptr(dev)), "mdg_cfconv_bwd_smear")
        return gW1, gb1, gW2, gmu, gcf
    fn = lib.mdg_cfconv_bwd_bf16 if bf16 else lib.mdg_cfconv_bwd
    check(fn(C.byref(fnet.struct), ptr(d), ptr(dd), ptr(topo.nbr), topo.n_edges, ptr(h), ptr(hd), ptr(mb),
             ptr(mdb), ptr(d_b), ptr(dd_b), ptr(gW1), ptr(gb1), ptr(gW2), ptr(ws), nv, stream_ptr(dev)), "mdg_cfconv_bwd")
    return (gW1, gb1, gW2) if want_theta else None


DENSE_MAX_K = 4096       # (k > 256 runs as k-slabs of 256 inside mdg_dense)


def dense(W, x0, trans=False, bias=None, act=False, mul=None, res=None, x1=None, res1=None, want_sig=False):
    """Node-level Dense with fused epilogue on the f32 MFMA (csrc/dense.hip):
    out0 = act(x0 B + bias) * mul + res ; out1 = act'(.) (x1 B) + res1 with B = W^T (trans False) or W (True).
    Returns (out0, sig0, out1)."""
    lib = _lib.load()
    require_gpu(x0, "x0")
    x0 = x0.contiguous()
    W = W.detach().contiguous()
    N, K = x0.shape
    M = W.shape[1] if trans else W.shape[0]
    assert (W.shape[0] if trans else W.shape[1]) == K, "dense: shape mismatch"
    dev = x0.device
    cont = lambda t: t.contiguous() if t is not None else None
    tops = _torch_ops.get()
    if tops is not None:
        out0, sig, out1 = tops.dense_ssp(W, bool(trans), bool(act), x0, cont(bias), cont(mul), cont(res), cont(x1), cont(res1),
                                         bool(want_sig))
        return out0, (sig if (act and want_sig) else None), (out1 if x1 is not None else None)
    out0 = torch.empty(N, M, device=dev)
    sig = torch.empty(N, M, device=dev) if (act and want_sig) else None
    out1 = None
    if x1 is not None:
        x1 = x1.contiguous()
        out1 = torch.empty(N, M, device=dev)
    cont = lambda t: t.contiguous() if t is not None else None
    check(lib.mdg_dense(ptr(W), int(bool(trans)), int(bool(act)), N, K, M, ptr(x0), ptr(cont(bias)), ptr(cont(mul)),
                        ptr(cont(res)), ptr(out0), ptr(sig), ptr(x1), ptr(cont(res1)), ptr(out1), stream_ptr(dev)),
          "mdg_dense")
    return out0, sig, out1


# ----------------------------------------------------------------------------- fused elementwise pieces
def smear(d, mu, c):
    lib = _lib.load()
    E, G = d.shape[0], mu.shape[0]
    g, phi = torch.empty(E, G, device=d.device), torch.empty(E, G, device=d.device)
    check(lib.mdg_smear(ptr(d.contiguous()), ptr(mu.contiguous()), ptr(c.contiguous()), E, G, ptr(g), ptr(phi),
                        stream_ptr(d.device)), "mdg_smear")
    return g, phi


def ssp(a, with_sigmoid=False):
    lib = _lib.load()
    a = a.contiguous()
    s = torch.empty_like(a)
    sa = torch.empty_like(a) if with_sigmoid else None
    check(lib.mdg_ssp(ptr(a), a.numel(), ptr(s), ptr(sa), stream_ptr(a.device)), "mdg_ssp")
    return (s, sa) if with_sigmoid else s


def mul_row(x, y, r=None):
    lib = _lib.load()
    x, y = x.contiguous(), y.contiguous()
    o = torch.empty_like(x)
    check(lib.mdg_mul_row(ptr(x), ptr(y), ptr(r.contiguous()) if r is not None else None, x.shape[0], x.shape[1],
                          ptr(o), stream_ptr(x.device)), "mdg_mul_row")
    return o


def ssp_dual_bwd(sa, xd, sdb, sb):
    lib = _lib.load()
    sa, xd, sdb, sb = sa.contiguous(), xd.contiguous(), sdb.contiguous(), sb.contiguous()
    xdb, xb = torch.empty_like(sa), torch.empty_like(sa)
    check(lib.mdg_ssp_dual_bwd(ptr(sa), ptr(xd), ptr(sdb), ptr(sb), sa.numel(), ptr(xdb), ptr(xb),
                               stream_ptr(sa.device)), "mdg_ssp_dual_bwd")
    return xdb, xb


def ssp_dual_bwd_t(sa, td, sdb, sb):
    """ssp_dual_bwd with the tangent stored as t_dot = sa * x_dot."""
    lib = _lib.load()
    sa, td, sdb, sb = sa.contiguous(), td.contiguous(), sdb.contiguous(), sb.contiguous()
    tops = _torch_ops.get()
    if tops is not None:
        return tuple(tops.ssp_dual_bwd_t(sa, td, sdb, sb))
    xdb, xb = torch.empty_like(sa), torch.empty_like(sa)
    check(lib.mdg_ssp_dual_bwd_t(ptr(sa), ptr(td), ptr(sdb), ptr(sb), sa.numel(), ptr(xdb), ptr(xb),
                                 stream_ptr(sa.device)), "mdg_ssp_dual_bwd_t")
    return xdb, xb


def readout_head(sy, syd, L2):
    """(ydb, yb) = (sy * L2, (1 - sy) * syd * L2): the seeds of the reverse sweep through the readout, one launch
    (csrc/elem.hip); syd None: first-order pass, yb None."""
    lib = _lib.load()
    require_gpu(sy, "sy")
    sy = sy.contiguous()
    syd = syd.contiguous() if syd is not None else None
    l2 = L2.detach().reshape(-1).contiguous()
    ydb = torch.empty_like(sy)
    yb = torch.empty_like(sy) if syd is not None else None
    check(lib.mdg_readout_head(ptr(sy), ptr(syd), ptr(l2), sy.shape[0], sy.shape[1], ptr(ydb), ptr(yb),
                               stream_ptr(sy.device)), "mdg_readout_head")
    return ydb, yb


class ThetaAccum:
    """A flat parameter-gradient buffer in `params` order (tinydiffeq.py:106-108) that kernels accumulate into directly:
    `flat[offset(p) ...] += alpha * (t[idx] - t[idx - 1]) * value` with the time grid and the frame index on the device
    (sovlers.py:160; t = idx = None: plain alpha)."""

    def __init__(self, params, flat=None, t=None, idx=None):
        self.params = list(params)
        self.off, n = {}, 0
        for p in self.params:
            self.off[id(p)] = n
            n += p.numel()
        self.n = n
        self.flat = flat if flat is not None else torch.zeros(n, device=self.params[0].device, dtype=torch.float32)
        assert self.flat.numel() == n and self.flat.is_contiguous()
        self.t, self.idx = t, idx

    def views(self):
        return [self.flat[self.off[id(p)]:self.off[id(p)] + p.numel()].view(p.shape) for p in self.params]


class GradJobs:
    """Collects the reductions of one adjoint evaluation and issues them as two launches per <= 32 jobs
    (mdg_grad_jobs, csrc/gradjobs.hip)."""

    def __init__(self):
        self.jobs, self.keep = [], []

    def _add(self, kind, off, rows, m, n, A, B=None, A2=None, B2=None, row_map=None):
        ts = [x.contiguous() if x is not None else None for x in (A, B, A2, B2)]
        self.keep.extend(ts)
        if row_map is not None:
            self.keep.append(row_map)
        self.jobs.append((kind, int(off), int(rows), int(m), int(n), ts, row_map))

    def atb(self, off, A, B, A2=None, B2=None, row_map=None):
        """flat[off ...] <- A^T B (+ A2^T B2), [m, n] row-major (rows re-mapped through row_map)."""
        self._add(_lib.GRAD_ATB, off, A.shape[0], A.shape[1], B.shape[1], A, B, A2, B2, row_map)

    def colsum(self, off, A, B=None, A2=None, B2=None):
        """flat[off ...] <- sum over rows of A (.* B) (+ A2 (.* B2))."""
        self._add(_lib.GRAD_COLSUM, off, A.shape[0], A.shape[1], 0, A, B, A2, B2)

    def axpy(self, off, A):
        """flat[off ...] <- A (already reduced)."""
        self._add(_lib.GRAD_AXPY, off, 1, A.numel(), 0, A.reshape(-1))

    def run(self, acc, alpha=1.0, accumulate=True):
        lib = _lib.load()
        dev = acc.flat.device
        for c0 in range(0, len(self.jobs), _lib.GRAD_JOBS_MAX):
            chunk = self.jobs[c0:c0 + _lib.GRAD_JOBS_MAX]
            arr = (_lib.MdgGradJob * len(chunk))()
            for j, (kind, off, rows, m, n, ts, row_map) in zip(arr, chunk):
                j.A, j.B, j.A2, j.B2 = (x.data_ptr() if x is not None else None for x in ts)
                j.row_map = row_map.data_ptr() if row_map is not None else None
                j.rows, j.m, j.n, j.kind, j.out_off = rows, m, n, kind, off
            need = int(lib.mdg_grad_jobs_workspace(arr, len(chunk)))
            if need < 0:
                check(1, "mdg_grad_jobs_workspace")
            ws = torch.empty(max(1, need), device=dev, dtype=torch.float32)
            self.keep.append(ws)
            check(lib.mdg_grad_jobs(arr, len(chunk), ptr(acc.flat), float(alpha), ptr(acc.t), ptr(acc.idx), int(bool(accumulate)),
                                    ptr(ws), stream_ptr(dev)), "mdg_grad_jobs")


class _ChainOut:
    __slots__ = ("out0", "out1", "sig", "pre0", "pre1", "out0_h", "out1_h")


class RowChain:
    """A chain of node-level Dense layers as ONE launch (mdg_row_chain, csrc/rowchain.hip): every stage reads the
    previous stage's outputs from on-chip memory (or `in0` / `in1`) and leaves global copies of its outputs in fresh
    [n_rows, M] tensors.  `dual`: primal + tangent rows (or the two adjoints) share the weights.  `x3`: MDG_CHAIN_X3, the
    products as three bf16 MFMAs on split operands (the rows16 precision option's companion, `chain_x3`)."""

    def __init__(self, n_rows, dual, device, x3=0):
        # x3: 0, or the flag MDG_CHAIN_X3 (2) / MDG_CHAIN_X6 (4) (True: X3)
        self.N, self.dual, self.dev, self.x3 = int(n_rows), bool(dual), device, (2 if x3 is True else int(x3))
        self.stages, self.keep, self.words = [], [], []

    @staticmethod
    def supported(*widths):
        return all(1 <= int(w) <= _lib.CHAIN_MAX_WIDTH for w in widths)

    def stage(self, W, trans=False, bias=None, act=False, mode=0, in0=None, in1=None, res0=None, res1=None, aux0=None,
              aux1=None, want_sig=False, want_pre=(False, False), store=True, mirror=False):
        """-> the stage's output tensors (out0, out1, sig, pre0, pre1; None where not produced).  mirror: bf16 copies
        out0_h / out1_h of the outputs as well (what the rows16 cfconv kernels gather); with store=False only they are
        written."""
        # (host cost matters: a stacked SchNet pass builds ~70 chains, and below ~1 ms of GPU work per MD step the launch
        #  thread is what the GPU waits for -- the descriptors are packed as plain integers, MdgChainStage's layout)
        assert len(self.stages) < _lib.CHAIN_MAX_STAGES, "row chain: too many stages"
        if not W.is_contiguous():
            W = W.contiguous()
        K, M = (W.shape[0], W.shape[1]) if trans else (W.shape[1], W.shape[0])
        N, dev, dual = self.N, self.dev, self.dual
        o = _ChainOut()
        f32, head = torch.float32, mode == _lib.CHAIN_HEAD
        o.out0 = torch.empty(N, M, device=dev, dtype=f32) if store else None
        o.out1 = torch.empty(N, M, device=dev, dtype=f32) if (store and dual) else None
        o.sig = torch.empty(N, M, device=dev, dtype=f32) if (act and want_sig) else None
        o.pre0 = torch.empty(N, M, device=dev, dtype=f32) if (head and want_pre[0]) else None
        o.pre1 = torch.empty(N, M, device=dev, dtype=f32) if (head and want_pre[1] and dual) else None
        o.out0_h = torch.empty(N, M, device=dev, dtype=torch.bfloat16) if mirror else None
        o.out1_h = torch.empty(N, M, device=dev, dtype=torch.bfloat16) if (mirror and dual) else None
        words, keep = self.words, self.keep
        words.append(W.data_ptr())
        keep.append(W)
        for t in (bias, in0, in1, res0, res1, aux0, aux1):
            if t is None:
                words.append(0)
            else:
                if not t.is_contiguous():
                    t = t.contiguous()
                keep.append(t)
                words.append(t.data_ptr())
        for t in (o.out0, o.out1, o.sig, o.pre0, o.pre1, o.out0_h, o.out1_h):
            words.append(0 if t is None else t.data_ptr())
        words.append(int(K) | (int(M) << 32))                       # int32 K, M
        words.append((1 if trans else 0) | ((1 if act else 0) << 32))   # int32 trans, act
        words.append(int(mode))                                     # int32 mode, pad_
        self.stages.append(o)
        return o

    def run(self):
        lib = _lib.load()
        buf = _array("Q", self.words)                               # [n_stages] MdgChainStage, 18 x 8 bytes each
        assert len(self.words) == 18 * len(self.stages) and C.sizeof(_lib.MdgChainStage) == 144
        check(lib.mdg_row_chain(buf.buffer_info()[0], len(self.stages), self.N, int(self.dual) | self.x3, stream_ptr(self.dev)),
              "mdg_row_chain")


def rows_to_bf16(x):
    """Round-to-nearest-even bf16 copy of an f32 node matrix [N, M] (M a multiple of 4): mdg_rows_to_bf16."""
    lib = _lib.load()
    require_gpu(x, "x")
    x = x.contiguous()
    out = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    check(lib.mdg_rows_to_bf16(ptr(x), x.shape[0], x.shape[1], x.shape[1], ptr(out), stream_ptr(x.device)), "mdg_rows_to_bf16")
    return out


def smear_bwd(gdb, gb, g, phi, dd, c, d_b, dd_b):
    """Accumulates into d_b (and dd_b when gdb is given) in place."""
    lib = _lib.load()
    gb = gb.contiguous()
    gdb = gdb.contiguous() if gdb is not None else None
    check(lib.mdg_smear_bwd(ptr(gdb), ptr(gb), ptr(g), ptr(phi), ptr(dd), ptr(c.contiguous()), g.shape[0], g.shape[1],
                            ptr(d_b), ptr(dd_b), stream_ptr(g.device)), "mdg_smear_bwd")
