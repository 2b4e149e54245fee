"""One C-ABI call per SchNet evaluation (`mdg_schnet_force` / `mdg_schnet_force_vjp`, csrc/schnet_eval.hip).

`analytic._force_chain` / `_force_vjp_chain` issue the 20-30 launches of an evaluation one ctypes call at a time, with a fresh
torch tensor for every intermediate; three evaluations make an MD step (torchmd/sovlers.py:106-168, 211-293), and on the
stacked 8 x 4 096-bead pass the launch thread kept only 0.05 ms ahead of the GPU.  A `SchnetPlan` describes the same network
once -- device pointers of the weights by role, their offsets in the flat parameter-gradient vector, the persistent
embedding / first-filter rows -- owns ONE workspace, and hands an evaluation to the C++ loop that enqueues the identical
launch sequence (same kernels, same operands: bitwise the same results, tests/test_gpu_schnet_plan.py).

Not taken (the launch-by-launch path runs instead): a trainable radial basis (its gradient is chained through torch ops),
networks outside the fused kernels / row chains, `SchNet.eval_plan = False`, MDG_SCHNET_PLAN=0.
"""
import ctypes as C
import os

import torch

from .. import _lib, ops
from .._lib import check, ptr, stream_ptr


def enabled(net):
    return getattr(net, "eval_plan", True) is not False and os.environ.get("MDG_SCHNET_PLAN", "1") != "0"


class SchnetPlan:
    """The plan of one network on one device; rebuilt when a tensor it points at is replaced or reallocated."""

    def __init__(self, net, Ps, fns):
        lib = _lib.load()
        if int(lib.mdg_schnet_plan_sizeof()) != C.sizeof(_lib.MdgSchnetPlan):
            raise RuntimeError("mdgrad_amd: MdgSchnetPlan layout differs between _lib.py and libmdgrad_hip.so")
        self.net = net
        ro = net.atomwisereadout.readout["energy"]
        self.L1, self.l1, self.L2, self.l2 = ro[0].weight, ro[0].bias, ro[2].weight, ro[2].bias
        self.Ps, self.fns = Ps, fns
        s = _lib.MdgSchnetPlan()
        s.n_layers = len(Ps)
        s.n_atom_basis = int(Ps[0]["Wn"].shape[1])
        s.n_readout = int(self.L1.shape[0])
        self.keep = []
        for i, (P, fn) in enumerate(zip(Ps, fns)):
            L = s.layer[i]
            L.filt = fn.struct
            for role in ("Wn", "bn", "U1", "c1", "U2", "c2"):
                t = P[role].detach()
                assert t.is_contiguous() and t.dtype == torch.float32
                setattr(L, role, t.data_ptr())
            L.bf16, L.bf16_rev = int(bool(fn.bf16)), int(bool(fn.bf16 and fn.bf16_reverse))
            L.rows16, L.b2col = int(bool(fn.rows16)), int(bool(fn.b2col))
            self.keep.append(fn)
        s.L1, s.l1, s.L2 = self.L1.data_ptr(), self.l1.data_ptr(), self.L2.data_ptr()
        # the G-wide filter stash (mdg_cfconv_filter_stash): same results as recomputing; MDG_SCHNET_STASH=0 for the A/B
        s.stash = int(getattr(net, "filter_stash", True) is not False and os.environ.get("MDG_SCHNET_STASH", "1") != "0")
        from . import analytic
        s.chain_x3 = int(analytic._chain_x3(net, fns))       # (the launch-by-launch path asks the same question: same kernels)
        self.struct = s
        self.key = self._key(Ps, fns)
        self.ws = {}                 # one workspace per evaluation kind (dual, theta): a captured HIP graph of one kind keeps
        self._retired = []           # its addresses whatever the others do; outgrown workspaces stay alive for old graphs
        self._off_for = None

    @staticmethod
    def _key(Ps, fns):
        k = []
        for P, fn in zip(Ps, fns):
            k.append(tuple(P[r].data_ptr() for r in ("Wn", "bn", "U1", "c1", "U2", "c2")) + (id(fn),))
        return tuple(k)

    def matches(self, Ps, fns):
        from . import analytic
        ro = self.net.atomwisereadout.readout["energy"]
        return (self.key == self._key(Ps, fns) and int(analytic._chain_x3(self.net, fns)) == int(self.struct.chain_x3) and ro[0].weight is self.L1 and ro[0].bias is self.l1 and ro[2].weight is self.L2
                and self.struct.L1 == self.L1.data_ptr() and self.struct.L2 == self.L2.data_ptr())

    # -- per call ---------------------------------------------------------------------------------
    def _topology(self, topo):
        s, e = self.struct, topo.ell
        s.n_atoms, s.n_edges = int(topo.n_atoms), int(topo.n_edges)
        s.nbr, s.offsets = topo.nbr.data_ptr(), topo.offsets.data_ptr()
        s.col, s.eid, s.cnt, s.max_nbr = e.col.data_ptr(), topo.eid.data_ptr(), e.cnt.data_ptr(), int(e.max_nbr)
        nv = getattr(topo, "n_valid", None)
        s.n_valid = nv.data_ptr() if nv is not None else None
        verlet = getattr(topo, "verlet", None)
        if verlet is not None:
            s.masked, s.cell, s.cutoff = 1, verlet[0], float(verlet[1])
        else:
            s.masked = 0

    def _rows(self, r0, h0, h0_16):
        s = self.struct
        s.r0, s.h0 = r0.data_ptr(), h0.data_ptr()
        s.h0_16 = h0_16.data_ptr() if h0_16 is not None else None
        self.keep_rows = (r0, h0, h0_16)

    def _offsets(self, acc, z):
        """Offsets of every parameter in `acc`'s flat buffer, and the species table of the embedding's gradient."""
        if self._off_for is not acc.off:
            s, net = self.struct, self.net
            off = lambda p: int(acc.off[id(p)])
            for i, conv in enumerate(net.convolutions):
                md = conv.moduledict
                L = s.layer[i]
                f, n, u = md["message_edge_filter"], md["message_node_filter"], md["update_function"]
                L.off_W1, L.off_b1, L.off_W2, L.off_b2 = off(f[1].weight), off(f[1].bias), off(f[3].weight), off(f[3].bias)
                L.off_Wn, L.off_bn = off(n.weight), off(n.bias)
                L.off_U1, L.off_c1, L.off_U2, L.off_c2 = off(u[0].weight), off(u[0].bias), off(u[2].weight), off(u[2].bias)
            ro = net.atomwisereadout.readout["energy"]
            s.off_L1, s.off_l1, s.off_L2 = off(ro[0].weight), off(ro[0].bias), off(ro[2].weight)
            s.off_embed = off(net.atom_embed.weight)
            self._off_for = acc.off
        from . import analytic
        uniq, onehot = analytic._species_onehot(z)
        s = self.struct
        s.onehot, s.uniq, s.n_species = onehot.data_ptr(), uniq.data_ptr(), int(uniq.shape[0])
        self.keep_species = (uniq, onehot)

    def _workspace(self, dual, theta, dev):
        lib = _lib.load()
        need = int(lib.mdg_schnet_workspace(C.byref(self.struct), int(dual), int(theta)))
        if need < 0:
            raise RuntimeError("mdgrad_amd: mdg_schnet_workspace rejected the plan")
        kind = (bool(dual), bool(theta))
        ws = self.ws.get(kind)
        if ws is None or ws.numel() < need or ws.device != dev:
            if ws is not None:
                self._retired.append(ws)
            ws = self.ws[kind] = torch.empty(need + need // 16 + 64, device=dev, dtype=torch.float32)
        self.struct.ws, self.struct.ws_floats = ws.data_ptr(), int(ws.numel())

    def _energy(self, want, dev):
        if not want:
            return None
        return torch.empty(int(self.struct.n_readout), device=dev, dtype=torch.float32)

    def _finish_energy(self, ecol, n_atoms):
        if ecol is None:
            return None
        U = (self.L2.detach() * ecol).sum()
        return U + n_atoms * self.l2.detach().sum() if self.l2 is not None else U

    def force(self, x, topo, rows, want_energy):
        lib = _lib.load()
        dev = x.device
        self._topology(topo)
        self._rows(*rows)
        self._workspace(False, False, dev)
        F = torch.empty(topo.n_atoms, 3, device=dev, dtype=torch.float32)
        ecol = self._energy(want_energy, dev)
        check(lib.mdg_schnet_force(C.byref(self.struct), ptr(x), ptr(F), ptr(ecol), stream_ptr(dev)), "mdg_schnet_force")
        return self._finish_energy(ecol, topo.n_atoms), F

    def force_vjp(self, x, w, topo, rows, z, acc, want_energy):
        """acc: ops.ThetaAccum (or None: no parameter gradients)."""
        lib = _lib.load()
        dev = x.device
        self._topology(topo)
        self._rows(*rows)
        if acc is not None:
            self._offsets(acc, z)
        self._workspace(True, acc is not None, dev)
        F = torch.empty(topo.n_atoms, 3, device=dev, dtype=torch.float32)
        dwf = torch.empty(topo.n_atoms, 3, device=dev, dtype=torch.float32)
        ecol = self._energy(want_energy, dev)
        check(lib.mdg_schnet_force_vjp(C.byref(self.struct), ptr(x), ptr(w), ptr(F), ptr(dwf),
                                       ptr(acc.flat) if acc is not None else None, -1.0,
                                       ptr(acc.t) if acc is not None else None, ptr(acc.idx) if acc is not None else None,
                                       ptr(ecol), stream_ptr(dev)), "mdg_schnet_force_vjp")
        return self._finish_energy(ecol, topo.n_atoms), F, dwf


def get(net, Ps, fns):
    """The network's plan (cached on it), or None when this network does not take one."""
    from . import analytic
    if not enabled(net):
        return None
    if any(analytic._trainable_smear(c) for c in net.convolutions):
        return None
    if len(Ps) > _lib.SCHNET_MAX_LAYERS or net.atomwisereadout.readout["energy"][2].weight.shape[0] != 1:
        return None
    ro = net.atomwisereadout.readout["energy"]
    if ro[0].bias is None or any(P[r] is None for P in Ps for r in ("bn", "c1", "c2")):
        return None
    p = net.__dict__.get("_mdg_plan")
    if p is None or not p.matches(Ps, fns):
        p = SchnetPlan(net, Ps, fns)
        net.__dict__["_mdg_plan"] = p
    return p
