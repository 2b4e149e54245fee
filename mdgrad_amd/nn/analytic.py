"""Hand-derived first- and second-order passes of the SchNet energy (no autograd graph).

The adjoint of the MD step needs, per evaluation, the force F = -dU/dx and, for a given atom
vector w (= lambda_v / m), the vjps d(w.F)/dx and d(w.F)/dtheta.  PyTorch gets them by
differentiating the energy twice (reverse-over-reverse through ~500 small ops).  Here the same
quantities come from ONE primal forward, ONE forward-mode (tangent) sweep along x_dot = w -- which
gives U_dot = dU/dx . w = -(w.F) -- and ONE reverse sweep of U_dot, all written out explicitly on
top of the HIP kernels:

    force(...)      primal forward + reverse of U                      (E1)
    force_vjp(...)  primal + tangent forward, reverse of U_dot          (E2)

The reverse sweep of U_dot already contains the reverse sweep of U: U_dot is linear in the tangents with the
partial derivatives of U as coefficients, so the adjoint of every tangent quantity IS the reverse-mode adjoint
of its primal (rdb = dU/dr, mdb = dU/dm, ..., dd_b = dU/dd) and the force -dU/dx falls out of the same sweep.

Edge level (everything of size [E, .]): the fused interaction-block kernels of csrc/cfconv_fused.hip -- filter
network on the MFMA, gather-multiply-sum per atom, and the adjoints with the filter recomputed -- so no
[E,G] / [E,F] tensor exists.  Node level ([N, .]): GEMMs + small fused elementwise kernels.  Shapes the fused
kernels do not take (n_gaussians > 64, n_filters not a multiple of 128 above 128 or above 512) run the unfused chain
(`_force_vjp_unfused`); a trainable radial basis (GaussianSmearing(trainable=True)) stays on the fused kernels, which then
also return the gradients of its centres and widths: graph
kernels of csrc/graph.hip, the split-K A^T B kernel and library GEMMs.

Notation follows SURVEY A.9 / nff/nn/models/schnet.py:113-171 (reference parameter names in
brackets): per layer  g = smear(d) -> a = W1 g + b1 [edge_filter.1] -> s = ssp(a) -> Wf = W2 s + b2
[edge_filter.3];  h = Wn r + bn [node_filter];  m = agg(h, Wf);  u = U1 m + c1 [update.0];
t = ssp(u);  r <- r + U2 t + c2 [update.2];  readout y = L1 r + l1, U = sum L2 ssp(y) + l2.
"""
import math
import os

import torch

from .. import ops, _lib as C_

_LN2 = math.log(2.0)


def _ssp(x):
    return torch.nn.functional.softplus(x) - _LN2


_warned = set()


def _warn_once(key, msg):
    """The library-GEMM fallbacks announce themselves once per process: nobody should time them as the
    product's hand-written path by accident."""
    if key not in _warned:
        _warned.add(key)
        import warnings
        warnings.warn("mdgrad_amd.nn.analytic: " + msg, RuntimeWarning, stacklevel=3)


def _addmm(bias, x, wt):
    """x @ wt + bias without the hipBLASLt epilogue path (see _blas_for)."""
    return x.mm(wt).add_(bias)


def _atb(a, b):
    """a^T b, tall-skinny aware."""
    if a.shape[0] >= ops.TALL_ROWS and a.is_cuda:
        return ops._atb(a, b)
    return a.t().mm(b)


def _atb_sum(a, b, c, d):
    """a^T b + c^T d (primal + tangent halves of one weight gradient; c may be None)."""
    if c is None:
        return _atb(a, b)
    if a.is_cuda and a.dtype == torch.float32 and a.shape[0] >= ops.TALL_ROWS:
        return ops._atb2(a, b, c, d)             # one split-K launch pair instead of two products and an add
    return a.t().mm(b) + c.t().mm(d)


_CAT_EDGES = 65536
_onehot_cache = {}


def _species_onehot(z):
    """(species present [S] int64, one-hot [N, S] fp32) of the atomic numbers, cached per index tensor
    (one host sync for `unique` the first time)."""
    key = (z.data_ptr(), z.shape[0], z.device)
    hit = _onehot_cache.get(key)
    if hit is None or hit[0] is not z:
        if len(_onehot_cache) > 16:
            _onehot_cache.clear()
        uniq, inv = torch.unique(z, return_inverse=True)
        hit = (z, uniq, torch.nn.functional.one_hot(inv, uniq.shape[0]).to(torch.float32).contiguous())
        _onehot_cache[key] = hit
    return hit[1], hit[2]


# The answers of supported / fused_ok / chain_ok depend on the network's STRUCTURE (module types, layer widths, a few
# switches), not on its weights; an MD pass asks them several times per force evaluation, and walking the module tree costs
# more host time than the launches it decides about (tools/hostprof_schnet.py).  They are cached on the network under a key
# that names everything they read: the switches, the conv modules' identities and their layer shapes.
def _conv_shape_key(conv):
    """What supported / fused_ok / chain_ok read of one interaction block beyond its identity: the layer shapes,
    whether the Gaussian basis is origin-centred and whether its width is trainable (a Parameter)."""
    md = conv._modules["moduledict"]._modules
    f, n, u = md["message_edge_filter"], md["message_node_filter"], md["update_function"]
    sm = f[0]
    return (id(conv), tuple(f[1].weight.shape), tuple(f[3].weight.shape), tuple(n.weight.shape), tuple(u[0].weight.shape),
            tuple(u[2].weight.shape), bool(getattr(sm, "centered", False)), isinstance(sm.width, torch.nn.Parameter))


def _structure_key(net):
    convs = getattr(net, "convolutions", None)
    if convs is None:
        return None
    ro = net.atomwisereadout
    try:
        ro_key = (id(ro), tuple(ro.readout.keys()), tuple(tuple(m.weight.shape) for m in ro.readout["energy"] if hasattr(m, "weight")),
                  ro.post_readout is None)
    except (AttributeError, KeyError, TypeError):
        ro_key = (id(ro),)
    return (getattr(net, "fused_block", True), getattr(net, "row_chain", True), os.environ.get("MDG_ROW_CHAIN", "1"),
            tuple(_conv_shape_key(c) for c in convs), ro_key)


def _cached(net, name, compute):
    key = _structure_key(net)
    if key is None:
        return compute(net)
    c = net.__dict__.get("_mdg_structure")
    if c is None or c[0] != key:
        c = (key, {})
        net.__dict__["_mdg_structure"] = c
    if name not in c[1]:
        c[1][name] = compute(net)
    return c[1][name]


def supported(net):
    return _cached(net, "supported", _supported)


def _supported(net):
    from .schnet import SchNet
    if not isinstance(net, SchNet) or list(net.atomwisereadout.readout.keys()) != ["energy"]:
        return False
    ro = net.atomwisereadout.readout["energy"]
    if len(ro) != 3 or net.atomwisereadout.post_readout is not None:
        return False
    for conv in net.convolutions:
        seq = conv.moduledict["message_edge_filter"]
        if seq[0].centered:
            return False                        # origin-centred Gaussians (not reachable through SchNetConv): autograd path
        if isinstance(seq[0].width, torch.nn.Parameter) and not (
                getattr(net, "fused_block", True) is not False
                and ops.FilterNet.supported(seq[0].offsets.shape[0], seq[3].weight.shape[0])):
            return False                        # a trainable basis is differentiated by the fused kernels only
    return True


def _trainable_smear(conv):
    return isinstance(conv.moduledict["message_edge_filter"][0].width, torch.nn.Parameter)


def _gauss_coeff(smear):
    """-0.5 / width^2 of a (`supported`) GaussianSmearing in a PERSISTENT buffer, refreshed in place when the width
    changed (three tiny launches per layer and evaluation otherwise).  A captured HIP graph holds the buffer's address:
    inside a capture the buffer is returned as it is, and the replaying pass refreshes every layer's buffer before its
    first replay (`refresh_embedding` through the interaction's `prepare_pass`), so an optimizer step on a trainable
    width reaches the replayed steps -- the same scheme as `_embedded` / `PairPotentials._theta`."""
    w = smear.width
    capturing = w.is_cuda and torch.cuda.is_current_stream_capturing()
    buf = getattr(smear, "_mdg_coeff_buf", None)
    if buf is None or buf.shape != w.shape or buf.device != w.device:
        if capturing:
            return (-0.5 / w.detach().pow(2)).contiguous()   # (no buffer yet: computed from the live width inside this graph)
        buf = smear._mdg_coeff_buf = torch.empty(w.shape, device=w.device, dtype=torch.float32)
        smear._mdg_coeff_key = None
    if capturing:
        return buf
    key = (w.data_ptr(), w._version)
    if smear._mdg_coeff_key != key:
        buf.copy_(-0.5 / w.detach().to(torch.float32).pow(2))
        smear._mdg_coeff_key = key
    return buf


def _layer_params(conv):
    """The tensors of one interaction block by role.  The parameter OBJECTS are looked up once per conv module (the module
    tree is walked ~25 times per block otherwise, per force evaluation); the cache is dropped when the block's submodules or
    one of their parameters were replaced.  `c` (the Gaussian coefficients) is refreshed on every call."""
    md = conv._modules["moduledict"]._modules
    f, n, u = md["message_edge_filter"], md["message_node_filter"], md["update_function"]
    c = conv.__dict__.get("_mdg_P")
    ok = c is not None and c[0] is f and c[1] is n and c[2] is u
    if ok:
        # every cached tensor is checked by identity against the module that owns it (a Parameter replaced by
        # assignment, parametrize or prune must reach the kernels AND keep its gradient slot) -- ten dictionary lookups
        for mod, attr, role in c[5]:
            if c[3][role] is not (mod._parameters.get(attr) if attr in mod._parameters else getattr(mod, attr)):
                ok = False
                break
    if not ok:
        slots = ((f[1], "weight", "W1"), (f[1], "bias", "b1"), (f[3], "weight", "W2"), (f[3], "bias", "b2"), (n, "weight", "Wn"),
                 (n, "bias", "bn"), (u[0], "weight", "U1"), (u[0], "bias", "c1"), (u[2], "weight", "U2"), (u[2], "bias", "c2"),
                 (f[0], "offsets", "mu"))
        P = {role: getattr(mod, attr) for mod, attr, role in slots}
        c = (f, n, u, P, f[0], slots)
        conv.__dict__["_mdg_P"] = c
    P = dict(c[3])
    P["c"] = _gauss_coeff(c[4])
    return P


@torch.no_grad()
def _primal(net, z, x, topo, offsets):
    delta = ops._edge_diff(x, topo) - offsets                 # schnet.py:142 (raw image flags by default)
    d = delta.pow(2).sum(1).sqrt()
    uhat = delta / d[:, None]
    r = net.atom_embed.weight[z]
    layers = []
    for conv in net.convolutions:
        P = _layer_params(conv)
        g, phi = ops.smear(d, P["mu"], P["c"])
        a = _addmm(P["b1"], g, P["W1"].t())
        s, sa = ops.ssp(a, True)
        Wf = _addmm(P["b2"], s, P["W2"].t())
        h = _addmm(P["bn"], r, P["Wn"].t())
        m = ops._cfconv_agg(h, Wf, topo)
        u = _addmm(P["c1"], m, P["U1"].t())
        t, su = ops.ssp(u, True)
        layers.append(dict(P=P, r=r, phi=phi, g=g, a=a, s=s, sa=sa, Wf=Wf, h=h, m=m, u=u, t=t, su=su))
        r = r + _addmm(P["c2"], t, P["U2"].t())
    ro = net.atomwisereadout.readout["energy"]
    L1, l1, L2, l2 = ro[0].weight, ro[0].bias, ro[2].weight, ro[2].bias
    y = _addmm(l1, r, L1.t())
    U = (_ssp(y).mm(L2.t()) + l2).sum()
    return dict(d=d, uhat=uhat, layers=layers, r=r, y=y, L1=L1, L2=L2, U=U)


@torch.no_grad()
def _reverse_U(fw, topo):
    """dU/dd per edge -> force."""
    rb = (torch.sigmoid(fw["y"]) * fw["L2"]).mm(fw["L1"])
    d_b = torch.zeros_like(fw["d"])
    for L in reversed(fw["layers"]):
        P = L["P"]
        ub = ops.mul_row(L["su"], rb.mm(P["U2"]))
        mb = ub.mm(P["U1"])
        hb = ops._cfconv_agg(mb, L["Wf"], topo)
        Wfb = ops._edge_prod(mb, L["h"], topo)
        rb = rb + hb.mm(P["Wn"])
        ab = ops.mul_row(L["sa"], Wfb.mm(P["W2"]))
        ops.smear_bwd(None, ab.mm(P["W1"]), L["g"], L["phi"], None, P["c"], d_b, None)
    return -ops._edge_scatter(d_b[:, None] * fw["uhat"], topo)


BUCKET_EDGES, BUCKET = 65536, 8192


def _stable(topo):
    """Above BUCKET_EDGES edges (GPU-bound) work on the bucket-padded topology so that GEMM shapes repeat."""
    if topo.n_edges >= BUCKET_EDGES and not getattr(topo, "padded", False):
        return topo.bucketed(BUCKET)
    return topo


class _blas_for:
    """Library-GEMM backend per evaluation (restored on exit).  rocBLAS ("hipblas") issues a GEMM in ~7 us
    where hipBLASLt needs ~19 us (tools/mmbench.py) -- what matters for small, launch-bound systems -- but on
    [E,128] operands with E ~ 1e5..1e6 the hipBLASLt kernels are ~2x faster (tools/mmbench2.py).  hipBLASLt
    searches its heuristics once per new shape (~1 ms per GEMM call, tools/prof_fwd.py), so it is only used
    when the edge count is padded to a repeating size (bucketed / fixed-capacity topologies)."""

    def __init__(self, topo):
        big = topo.n_edges >= 16384 and getattr(topo, "padded", False)
        self.want = "hipblaslt" if big else "hipblas"
        self.prev = None

    def __enter__(self):
        if self.want is not None and torch.cuda.is_available():
            try:
                self.prev = torch.backends.cuda.preferred_blas_library()
                torch.backends.cuda.preferred_blas_library(self.want)
            except Exception:
                self.prev = None
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.backends.cuda.preferred_blas_library(self.prev)
        return False


def fused_ok(net):
    """The fused interaction-block kernels take every layer of this network (and are not switched off)."""
    return _cached(net, "fused_ok", _fused_ok)


def _fused_ok(net):
    if getattr(net, "fused_block", True) is False:
        return False
    for conv in net.convolutions:
        seq = conv.moduledict["message_edge_filter"]
        if not ops.FilterNet.supported(seq[0].offsets.shape[0], seq[3].weight.shape[0]):
            return False
    return True


class _node_blas:
    """Node-level GEMMs ([N, A..F] operands) go through rocBLAS: its launch costs ~7 us against hipBLASLt's
    ~19 us, and the shapes are small (see _blas_for)."""

    def __enter__(self):
        self.prev = None
        if torch.cuda.is_available():
            try:
                self.prev = torch.backends.cuda.preferred_blas_library()
                torch.backends.cuda.preferred_blas_library("hipblas")
            except Exception:
                self.prev = None
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.backends.cuda.preferred_blas_library(self.prev)
        return False


def _dense(W, x0, trans=False, bias=None, act=False, mul=None, res=None, x1=None, res1=None, want_sig=False):
    """(out0, sig0, out1) of ops.dense -- the MFMA node kernel with fused bias / shifted softplus / product /
    residual, one launch for a primal row and its tangent (or both adjoints); layers wider than its LDS-resident
    weight chunk (k > 256) go through the library GEMM with the same epilogue in torch ops."""
    if x0.shape[1] <= ops.DENSE_MAX_K and x0.is_cuda:
        return ops.dense(W, x0, trans, bias, act, mul, res, x1, res1, want_sig)
    _warn_once("dense_k", "a node-level layer with %d input features (> %d) runs on the library GEMM (rocBLAS / hipBLASLt) with "
               "torch epilogues, not on the MFMA node kernel of csrc/dense.hip" % (x0.shape[1], ops.DENSE_MAX_K))
    B = W if trans else W.t()
    z0 = x0.mm(B)
    if bias is not None:
        z0 = z0 + bias
    z1 = x1.mm(B) if x1 is not None else None
    sig = None
    if act:
        sig = torch.sigmoid(z0)
        z0 = _ssp(z0)
        z1 = sig * z1 if z1 is not None else None
    if mul is not None:
        z0 = z0 * mul
    if res is not None:
        z0 = z0 + res
    if res1 is not None and z1 is not None:
        z1 = z1 + res1
    return z0, sig, z1


@torch.no_grad()
def _embedded(net, z):
    """atom_embed.weight[z] in a persistent buffer, refreshed in place when the weights (an optimizer step bumps their
    version) or the species change: a trajectory evaluates it three times per MD step on the same inputs.  Inside a
    HIP-graph capture the buffer is returned as it is -- the replaying pass refreshes it once before its first replay
    (`refresh_embedding`, called through the interaction's `prepare_pass`), so the captured steps carry no gather."""
    wt = net.atom_embed.weight
    buf = getattr(net, "_embed_buf", None)
    shape = (z.shape[0], wt.shape[1])
    if buf is None or tuple(buf.shape) != shape or buf.device != wt.device:
        if wt.is_cuda and torch.cuda.is_current_stream_capturing():
            return wt.detach()[z]                # (no buffer yet: the gather becomes part of this graph)
        buf = net._embed_buf = torch.empty(shape, device=wt.device, dtype=wt.dtype)
        net._embed_key = None
    if wt.is_cuda and torch.cuda.is_current_stream_capturing():
        return buf
    key = (wt.data_ptr(), wt._version, z.data_ptr(), z._version, tuple(z.shape))
    if getattr(net, "_embed_key", None) != key:
        torch.index_select(wt.detach(), 0, z, out=buf)
        net._embed_key = key
    return buf


@torch.no_grad()
def _first_filter(net, z, P0):
    """(r, h) of the first interaction block: the embedding rows and h = message_node_filter(r) = Wn r + bn
    (nff/nn/modules.py:514-533).  Neither depends on the positions, so h lives in a persistent buffer like the embedding
    rows and is recomputed when a weight it reads changes; inside a HIP-graph capture the buffer is returned as it is and
    the replaying pass refreshes it first (`refresh_embedding`)."""
    r = _embedded(net, z)
    Wn, bn = P0["Wn"], P0["bn"]
    buf = getattr(net, "_h0_buf", None)
    shape = (z.shape[0], Wn.shape[0])
    capturing = r.is_cuda and torch.cuda.is_current_stream_capturing()
    if buf is None or tuple(buf.shape) != shape or buf.device != r.device:
        if capturing:
            return r, _dense(Wn, r, bias=bn)[0]      # (no buffer yet: the layer becomes part of this graph)
        buf = net._h0_buf = torch.empty(shape, device=r.device, dtype=r.dtype)
        net._h0_key = None
    if capturing:
        return r, buf
    key = (getattr(net, "_embed_key", None), Wn.data_ptr(), Wn._version, bn.data_ptr() if bn is not None else 0,
           bn._version if bn is not None else 0)
    if getattr(net, "_h0_key", None) != key:
        buf.copy_(_dense(Wn, r, bias=bn)[0])
        net._h0_key = key
    return r, buf


def _filter_net(conv, P, bf16, rows16):
    """ops.FilterNet of one block, rebuilt only when a tensor it points at was reallocated (its constructor is six
    detach / contiguous calls and a ctypes struct: ~15 us, twice per force evaluation)."""
    ts = [P[k] for k in ("mu", "c", "W1", "b1", "W2", "b2")]
    if not all(t.dtype == torch.float32 and t.is_contiguous() for t in ts):
        # (FilterNet would hold converted COPIES of these tensors: nothing to key a cache on)
        return ops.FilterNet(*ts, bf16=bf16, rows16=rows16)
    key = (bool(bf16), bool(rows16), bool(ops.FilterNet.bf16_reverse)) + tuple(t.data_ptr() for t in ts)
    c = conv.__dict__.get("_mdg_fn")
    if c is None or c[0] != key:
        c = (key, ops.FilterNet(P["mu"], P["c"], P["W1"], P["b1"], P["W2"], P["b2"], bf16=bf16, rows16=rows16))
        conv.__dict__["_mdg_fn"] = c
    return c[1]


def _chain_x3(net, fns):
    """The flag the node-level chains of this network hand to mdg_row_chain (csrc/rowchain.hip) for their Dense products:
    MDG_CHAIN_X3 (2) with the rows16 precision option -- three bf16 MFMAs on head / remainder splits, ~1e-5 per product, the
    gathered rows being rounded at 2^-9 anyway (`SchNet.chain_x3 = False` / MDG_CHAIN_X3=0: not) --, otherwise MDG_CHAIN_X6 (4):
    three EXACT bf16 pieces per operand and six piece products, the accuracy of the f32 matrix instruction at 3/8 of its issue
    time -- an OPT-IN (`SchNet.chain_x6 = True`, or MDG_CHAIN_X6=1): measured 1-4 % slower in the pass than the f32 instruction (three
    planes of weight fragments per lane cost the occupancy the matrix cycles saved, profiles/r06_chain_x6_ab.txt)."""
    import os
    if (all(fn.rows16 for fn in fns) and getattr(net, "chain_x3", True) is not False
            and os.environ.get("MDG_CHAIN_X3", "1") != "0"):
        return 2
    if getattr(net, "chain_x6", False) or os.environ.get("MDG_CHAIN_X6", "0") == "1":
        return 4
    return 0


def _rows16(net, conv):
    """Does this block's convolution read bf16 mirrors of its gathered node matrices (SchNet.node_rows_bf16)?"""
    if not getattr(net, "node_rows_bf16", False):
        return False
    if not (getattr(conv, "filter_bf16", False) or getattr(net, "filter_bf16", False)):
        return False
    P = _layer_params(conv)
    return bool(C_.load().mdg_cfconv_rows16_supported(int(P["mu"].shape[0]), int(P["W2"].shape[0])))


def _h0_mirror(net, h):
    """bf16 mirror of the first block's filtered rows (`_first_filter`): persistent like them, refreshed with them."""
    buf = getattr(net, "_h0_buf16", None)
    capturing = h.is_cuda and torch.cuda.is_current_stream_capturing()
    if buf is None or buf.shape != h.shape or buf.device != h.device:
        if capturing or h is not getattr(net, "_h0_buf", None):
            return ops.rows_to_bf16(h)               # (no buffer yet: the conversion becomes part of this graph)
        buf = net._h0_buf16 = torch.empty(h.shape, device=h.device, dtype=torch.bfloat16)
        net._h0_key16 = None
    if capturing:
        return buf
    if h is not getattr(net, "_h0_buf", None):
        return ops.rows_to_bf16(h)
    if getattr(net, "_h0_key16", None) != net._h0_key:
        buf.copy_(h)                                 # (round to nearest even, as mdg_rows_to_bf16)
        net._h0_key16 = net._h0_key
    return buf


def refresh_embedding(net, z):
    """Bring the persistent embedding rows (and the first block's filtered rows) up to date, before the graph replays
    of a pass."""
    for conv in net.convolutions:                                 # Gaussian coefficients of every layer (trainable widths)
        _gauss_coeff(conv.moduledict["message_edge_filter"][0])
    if fused_ok(net) and chain_ok(net):
        _, h0 = _first_filter(net, z, _layer_params(net.convolutions[0]))
        if _rows16(net, net.convolutions[0]):
            _h0_mirror(net, h0)
    else:
        _embedded(net, z)


def _forward_fused(net, z, x, topo, w=None, want_sums=False, want_energy=True):
    """Primal (and, with w, forward-mode tangent along x_dot = w) sweep on the fused kernels."""
    d, uhat, dd, ddel = ops.edge_geom(x, topo, w)
    r, rd = _embedded(net, z), None                               # r_dot^0 = 0
    layers = []
    for conv in net.convolutions:
        P = _layer_params(conv)
        fn = ops.FilterNet(P["mu"], P["c"], P["W1"], P["b1"], P["W2"], P["b2"],
                           bf16=getattr(conv, "filter_bf16", False) or getattr(net, "filter_bf16", False))
        h, _, hd = _dense(P["Wn"], r, bias=P["bn"], x1=rd)                       # message_node_filter (+ tangent)
        m, md, hsum, hdsum = ops.cfconv_fwd(fn, d, dd, h, hd, topo, want_sums)
        # update MLP: t = ssp(U1 m + c1), su = sigmoid(.), td = su * (U1 md)
        t, su, td = _dense(P["U1"], m, bias=P["c1"], act=True, x1=md, want_sig=True)
        layers.append(dict(P=P, fn=fn, r=r, rd=rd, h=h, hd=hd, m=m, md=md, hsum=hsum, hdsum=hdsum, t=t, su=su, td=td))
        r, _, rd = _dense(P["U2"], t, bias=P["c2"], res=r, x1=td, res1=rd)       # residual (schnet.py:149-151)
    ro = net.atomwisereadout.readout["energy"]
    L1, l1, L2, l2 = ro[0].weight, ro[0].bias, ro[2].weight, ro[2].bias
    # readout layer with its activation in the epilogue: ssp(y), sy = sigmoid(y), syd = sy * y_dot -- all the reverse
    # sweeps need of it
    ty, sy, syd = _dense(L1, r, bias=l1, act=True, x1=rd, want_sig=True)
    U = (ty.mm(L2.t()) + l2).sum() if want_energy else None         # (the integrators only ask for forces)
    return dict(d=d, uhat=uhat, dd=dd, ddel=ddel, layers=layers, r=r, rd=rd, sy=sy, syd=syd, L1=L1, L2=L2, U=U)


@torch.no_grad()
def _force_fused(net, z, x, topo, want_energy=True):
    fw = _forward_fused(net, z, x, topo, want_energy=want_energy)
    d = fw["d"]
    rb = _dense(fw["L1"], ops.readout_head(fw["sy"], None, fw["L2"])[0], trans=True)[0]
    dU_dd = torch.zeros_like(d)
    for idx in range(len(fw["layers"]) - 1, -1, -1):
        L = fw["layers"][idx]
        P = L["P"]
        ub = _dense(P["U2"], rb, trans=True, mul=L["su"])[0]
        mb = _dense(P["U1"], ub, trans=True)[0]
        ops.cfconv_bwd(L["fn"], d, None, topo, L["h"], None, None, mb, None, dU_dd)
        if idx > 0:                                               # (the embedding below layer 0 is not needed)
            hb = ops.cfconv_fwd(L["fn"], d, None, mb, None, topo)[0]
            rb = _dense(P["Wn"], hb, trans=True, res=rb)[0]
    F, _ = ops.edge_geom_bwd(None, dU_dd, None, None, fw["uhat"], None, topo)
    return fw["U"], F


@torch.no_grad()
def _force_vjp_fused(net, z, x, w, topo, want_theta=True, want_energy=True, accum=None):
    """`accum` (ops.ThetaAccum): the parameter gradients are accumulated into its flat buffer -- every reduction of the
    sweep in one batched launch pair (csrc/gradjobs.hip) -- and None is returned in their place; without it they come
    back as a list aligned with net.parameters()."""
    fw = _forward_fused(net, z, x, topo, w, want_sums=want_theta, want_energy=want_energy)
    d, dd = fw["d"], fw["dd"]
    L1, L2, rd = fw["L1"], fw["L2"], fw["rd"]
    # ---------------- reverse sweep of U_dot = sum_i L2 . (sig(y_i) * yd_i)
    ydb, yb = ops.readout_head(fw["sy"], fw["syd"], L2)
    ro = net.atomwisereadout.readout["energy"]
    jobs = acc = None
    if want_theta:
        acc = accum if accum is not None else ops.ThetaAccum(net.parameters())
        off = lambda p: acc.off[id(p)]
        jobs = ops.GradJobs()
        jobs.colsum(off(ro[2].weight), fw["syd"])                                    # (ro[2].bias: U_dot does not see it)
        jobs.atb(off(ro[0].weight), yb, fw["r"], ydb, rd)
        jobs.colsum(off(ro[0].bias), yb)
    rdb, _, rb = _dense(L1, ydb, trans=True, x1=yb)
    both = torch.zeros(2, d.shape[0], device=d.device, dtype=d.dtype)        # (one fill for the two per-edge accumulators)
    d_b, dd_b = both[0], both[1]
    convs = list(net.convolutions)
    for idx in range(len(convs) - 1, -1, -1):
        L, md_ = fw["layers"][idx], convs[idx].moduledict
        P = L["P"]
        tdb, _, tb = _dense(P["U2"], rdb, trans=True, x1=rb)
        if want_theta:
            jobs.atb(off(md_["update_function"][2].weight), rb, L["t"], rdb, L["td"])
            jobs.colsum(off(md_["update_function"][2].bias), rb)
        udb, ub = ops.ssp_dual_bwd_t(L["su"], L["td"], tdb, tb)
        mdb, _, mb = _dense(P["U1"], udb, trans=True, x1=ub)
        if want_theta:
            jobs.atb(off(md_["update_function"][0].weight), udb, L["md"], ub, L["m"])
            jobs.colsum(off(md_["update_function"][0].bias), ub)
        smear_t = want_theta and _trainable_smear(convs[idx])
        th = ops.cfconv_bwd(L["fn"], d, dd, topo, L["h"], L["hd"], mb, mdb, d_b, dd_b, want_theta, want_smear=smear_t)
        if smear_t:
            # GaussianSmearing(trainable=True), layers.py:34-83: offsets = the centres; coef = -0.5 / width^2 per Gaussian
            sm = md_["message_edge_filter"][0]
            jobs.axpy(off(sm.offsets), th[3])
            jobs.axpy(off(sm.width), th[4] * sm.width.detach().pow(-3))
        if want_theta:
            jobs.axpy(off(md_["message_edge_filter"][1].weight), th[0])
            jobs.axpy(off(md_["message_edge_filter"][1].bias), th[1])
            jobs.axpy(off(md_["message_edge_filter"][3].weight), th[2])
            # sum_e W_b[e] = sum_n mb_n (.) sum_{j in nbr(n)} h_j  (+ the tangent half)
            if L["hdsum"] is not None:
                jobs.colsum(off(md_["message_edge_filter"][3].bias), mb, L["hsum"], mdb, L["hdsum"])
            else:
                jobs.colsum(off(md_["message_edge_filter"][3].bias), mb, L["hsum"])
        if want_theta or idx > 0:
            # the aggregation is symmetric in the adjacency: fed (mdb, mb) the forward kernel returns the
            # adjoints (hdb, hb) of (hd, h)
            hdb, hb, _, _ = ops.cfconv_fwd(L["fn"], d, dd, mdb, mb, topo)
            if want_theta:
                if L["rd"] is not None:
                    jobs.atb(off(md_["message_node_filter"].weight), hb, L["r"], hdb, L["rd"])
                else:
                    jobs.atb(off(md_["message_node_filter"].weight), hb, L["r"])
                jobs.colsum(off(md_["message_node_filter"].bias), hb)
            rdb, _, rb = _dense(P["Wn"], hdb, trans=True, res=rdb, x1=hb, res1=rb)
    # dd_b = dU/dd (see the module docstring): force and d(w.F)/dx from one scatter
    F, dwf = ops.edge_geom_bwd(d_b, dd_b, d, dd, fw["uhat"], fw["ddel"], topo)
    if not want_theta:
        return fw["U"], F, dwf, None
    # embedding rows: one-hot(z)^T rb, row s of the product -> row uniq[s] of the table (no float atomics: index_add_ on a
    # handful of species serialises and is not reproducible)
    uniq, onehot = _species_onehot(z)
    jobs.atb(off(net.atom_embed.weight), onehot, rb, row_map=uniq)
    jobs.run(acc, alpha=-1.0, accumulate=True)                   # w.F = -U_dot
    if accum is not None:
        return fw["U"], F, dwf, None
    return fw["U"], F, dwf, acc.views()


# ------------------------------------------------------------------------------------------------ row chains
# Between two aggregations every layer is local to an atom's feature row: the launches of _force_fused /
# _force_vjp_fused that are not cfconv kernels collapse into one mdg_row_chain launch per stretch (csrc/rowchain.hip):
#   forward   [m -> update MLP -> residual -> next block's message_node_filter]                      per inner block
#   turn      [m -> update MLP -> residual -> readout -> head -> readout^T -> U2^T -> ssp' -> U1^T] last block
#   reverse   [hb -> message_node_filter^T + residual -> U2^T -> ssp' -> U1^T]                     per inner block
def chain_ok(net):
    """The row-chain kernel takes every node-level layer of this network (and is not switched off)."""
    return _cached(net, "chain_ok", _chain_ok)


def _chain_ok(net):
    if getattr(net, "row_chain", True) is False or os.environ.get("MDG_ROW_CHAIN", "1") == "0" or not fused_ok(net):
        return False
    ws = []
    for conv in net.convolutions:
        P = _layer_params(conv)
        ws += list(P["Wn"].shape) + list(P["U1"].shape) + list(P["U2"].shape)
    ws += list(net.atomwisereadout.readout["energy"][0].weight.shape)
    return ops.RowChain.supported(*ws)


class _FlatAcc:
    """What mdg_grad_jobs needs of an accumulator: a flat f32 buffer, no interval weights."""

    def __init__(self, n, dev):
        self.flat, self.t, self.idx = torch.empty(n, device=dev, dtype=torch.float32), None, None


def _head_energy(act, L2, l2):
    """sum over atoms of (L2 . act_n + l2) for the readout's last layer (nff/nn/utils.py:56-75, one output)."""
    acc = _FlatAcc(act.shape[1], act.device)
    jobs = ops.GradJobs()
    jobs.colsum(0, act)
    jobs.run(acc, alpha=1.0, accumulate=False)
    U = (L2.detach() * acc.flat).sum()                              # ([out, A / 2] * [A / 2]: every output row, as .sum() of the layer)
    return U + act.shape[0] * l2.detach().sum() if l2 is not None else U


def _chain_forward(net, z, x, topo, w, want_sums, want_energy):
    """Primal (+ tangent along w) sweep and the turn at the readout: -> (fw, adjoints entering the last block's
    aggregation).  With w = None: first order (the force), one row per atom instead of a dual pair."""
    dual = w is not None
    d, uhat, dd, ddel = ops.edge_geom(x, topo, w)
    convs = list(net.convolutions)
    Ps = [_layer_params(c) for c in convs]
    fns = [_filter_net(c, P, getattr(c, "filter_bf16", False) or getattr(net, "filter_bf16", False),
                       getattr(net, "node_rows_bf16", False)) for c, P in zip(convs, Ps)]
    ro = net.atomwisereadout.readout["energy"]
    L1, l1, L2, l2 = ro[0].weight, ro[0].bias, ro[2].weight, ro[2].bias
    N, dev = z.shape[0], x.device
    x3 = _chain_x3(net, fns)
    (r, h), rd, hd = _first_filter(net, z, Ps[0]), None, None      # r_dot^0 = 0; message_node_filter of the first block
    # rows16 blocks gather bf16 mirrors (hg, hgd) of (h, hd); the chain stage that produces a block's rows writes them
    hg, hgd = (_h0_mirror(net, h) if fns[0].rows16 else h), None
    layers, turn = [], None
    for i, P in enumerate(Ps):
        # (want_sums: the neighbour sums of the node rows, which the gradient of the second filter layer's bias is made of --
        #  not needed where the reverse sweep hands that gradient out itself, FilterNet.b2col)
        m, md, hsum, hdsum = ops.cfconv_fwd(fns[i], d, dd if dual else None, hg, hgd, topo, want_sums and not fns[i].b2col)
        ch = ops.RowChain(N, dual, dev, x3)
        a = ch.stage(P["U1"], bias=P["c1"], act=True, in0=m, in1=md, want_sig=True)         # t, su, td
        b = ch.stage(P["U2"], bias=P["c2"], res0=r, res1=rd)                                # residual (schnet.py:149-151)
        layers.append(dict(P=P, fn=fns[i], r=r, rd=rd, h=hg, hd=hgd, m=m, md=md, hsum=hsum, hdsum=hdsum, t=a.out0, su=a.sig,
                           td=a.out1))                            # (h / hd: what this block's kernels gather)
        r, rd = b.out0, b.out1
        if i + 1 < len(Ps):
            # (a rows16 block reads its filtered rows through the mirrors only: the f32 copies are not written)
            c = ch.stage(Ps[i + 1]["Wn"], bias=Ps[i + 1]["bn"], mirror=fns[i + 1].rows16, store=not fns[i + 1].rows16)
            h, hd = c.out0, c.out1
            hg, hgd = (c.out0_h, c.out1_h) if fns[i + 1].rows16 else (h, hd)
        else:
            y = ch.stage(L1, bias=l1, act=True, mode=C_.CHAIN_HEAD, aux0=L2, want_sig=True, want_pre=(want_energy, True))
            g = ch.stage(L1, trans=True)                                                    # rdb, rb
            if dual:
                e = ch.stage(P["U2"], trans=True, mode=C_.CHAIN_SSP_BWD, aux0=a.sig, aux1=a.out1)   # udb, ub
            else:
                e = ch.stage(P["U2"], trans=True, mode=C_.CHAIN_MUL, aux0=a.sig)
            f = ch.stage(P["U1"], trans=True, mirror=fns[i].rows16)                         # mdb, mb
            turn = dict(y=y, g=g, e=e, f=f)
        ch.run()
    y = turn["y"]
    U = None                                                        # (the integrators only ask for forces)
    if want_energy:
        # U = sum_n (L2 . ssp(y_n) + l2): the column sums of the head stage's activations on the reduction kernel of
        # csrc/gradjobs.hip (fixed order), then a dot product over A / 2 numbers -- no library GEMM on the chained path
        U = _head_energy(y.pre0, L2, l2)
    fw = dict(d=d, uhat=uhat, dd=dd, ddel=ddel, layers=layers, r=r, rd=rd, sy=y.sig, syd=y.pre1, L1=L1, L2=L2, U=U, x3=x3)
    return fw, turn


@torch.no_grad()
def _force_chain(net, z, x, topo, want_energy=True):
    fw, turn = _chain_forward(net, z, x, topo, None, False, want_energy)
    d, layers = fw["d"], fw["layers"]
    rb, mb = turn["g"].out0, turn["f"].out0
    mg = turn["f"].out0_h if layers[-1]["fn"].rows16 else mb     # (what the block's kernels gather: the bf16 mirror or mb itself)
    dU_dd = torch.zeros_like(d)
    for idx in range(len(layers) - 1, -1, -1):
        L = layers[idx]
        ops.cfconv_bwd(L["fn"], d, None, topo, L["h"], None, None, mg, None, dU_dd)
        if idx > 0:                                               # (the embedding below layer 0 is not needed)
            hb = ops.cfconv_fwd(L["fn"], d, None, mg, None, topo)[0]
            Lp = layers[idx - 1]
            ch = ops.RowChain(z.shape[0], False, x.device, fw["x3"])
            g = ch.stage(L["P"]["Wn"], trans=True, in0=hb, res0=rb)
            ch.stage(Lp["P"]["U2"], trans=True, mode=C_.CHAIN_MUL, aux0=Lp["su"], store=False)
            f = ch.stage(Lp["P"]["U1"], trans=True, mirror=Lp["fn"].rows16)
            ch.run()
            rb, mb = g.out0, f.out0
            mg = f.out0_h if Lp["fn"].rows16 else mb
    F, _ = ops.edge_geom_bwd(None, dU_dd, None, None, fw["uhat"], None, topo)
    return fw["U"], F


@torch.no_grad()
def _force_vjp_chain(net, z, x, w, topo, want_theta=True, want_energy=True, accum=None):
    """_force_vjp_fused with the node-level layers chained (same results, same argument meaning)."""
    fw, turn = _chain_forward(net, z, x, topo, w, want_theta, want_energy)
    d, dd, layers = fw["d"], fw["dd"], fw["layers"]
    ro = net.atomwisereadout.readout["energy"]
    ydb, yb = turn["y"].out0, turn["y"].out1
    rdb, rb = turn["g"].out0, turn["g"].out1
    udb, ub = turn["e"].out0, turn["e"].out1
    mdb, mb = turn["f"].out0, turn["f"].out1
    r16 = layers[-1]["fn"].rows16                                 # (mdg, mg: what the block's kernels gather -- bf16 mirrors or the rows)
    mdg, mg = (turn["f"].out0_h, turn["f"].out1_h) if r16 else (mdb, mb)
    jobs = acc = None
    if want_theta:
        acc = accum if accum is not None else ops.ThetaAccum(net.parameters())
        off = lambda p: acc.off[id(p)]
        jobs = ops.GradJobs()
        jobs.colsum(off(ro[2].weight), fw["syd"])                                    # (ro[2].bias: U_dot does not see it)
        jobs.atb(off(ro[0].weight), yb, fw["r"], ydb, fw["rd"])
        jobs.colsum(off(ro[0].bias), yb)
    both = torch.zeros(2, d.shape[0], device=d.device, dtype=d.dtype)        # (one fill for the two per-edge accumulators)
    d_b, dd_b = both[0], both[1]
    convs = list(net.convolutions)
    for idx in range(len(convs) - 1, -1, -1):
        L, md_ = layers[idx], convs[idx].moduledict
        P = L["P"]
        if want_theta:
            jobs.atb(off(md_["update_function"][2].weight), rb, L["t"], rdb, L["td"])
            jobs.colsum(off(md_["update_function"][2].bias), rb)
            jobs.atb(off(md_["update_function"][0].weight), udb, L["md"], ub, L["m"])
            jobs.colsum(off(md_["update_function"][0].bias), ub)
        smear_t = want_theta and _trainable_smear(convs[idx])
        b2col = want_theta and L["fn"].b2col
        th = ops.cfconv_bwd(L["fn"], d, dd, topo, L["h"], L["hd"], mg, mdg, d_b, dd_b, want_theta, want_smear=smear_t, want_b2=b2col)
        if smear_t:
            sm = md_["message_edge_filter"][0]
            jobs.axpy(off(sm.offsets), th[3])
            jobs.axpy(off(sm.width), th[4] * sm.width.detach().pow(-3))
        if want_theta:
            jobs.axpy(off(md_["message_edge_filter"][1].weight), th[0])
            jobs.axpy(off(md_["message_edge_filter"][1].bias), th[1])
            jobs.axpy(off(md_["message_edge_filter"][3].weight), th[2])
            if b2col:
                jobs.axpy(off(md_["message_edge_filter"][3].bias), th[-1])
            elif L["hdsum"] is not None:
                jobs.colsum(off(md_["message_edge_filter"][3].bias), mb, L["hsum"], mdb, L["hdsum"])
            else:
                jobs.colsum(off(md_["message_edge_filter"][3].bias), mb, L["hsum"])
        if want_theta or idx > 0:
            hdb, hb, _, _ = ops.cfconv_fwd(L["fn"], d, dd, mdg, mg, topo)
            if want_theta:
                if L["rd"] is not None:
                    jobs.atb(off(md_["message_node_filter"].weight), hb, L["r"], hdb, L["rd"])
                else:
                    jobs.atb(off(md_["message_node_filter"].weight), hb, L["r"])
                jobs.colsum(off(md_["message_node_filter"].bias), hb)
            if idx > 0:
                Lp = layers[idx - 1]
                ch = ops.RowChain(z.shape[0], True, x.device, fw["x3"])
                g = ch.stage(P["Wn"], trans=True, in0=hdb, in1=hb, res0=rdb, res1=rb)
                e = ch.stage(Lp["P"]["U2"], trans=True, mode=C_.CHAIN_SSP_BWD, aux0=Lp["su"], aux1=Lp["td"])
                f = ch.stage(Lp["P"]["U1"], trans=True, mirror=Lp["fn"].rows16)
                ch.run()
                rdb, rb, udb, ub, mdb, mb = g.out0, g.out1, e.out0, e.out1, f.out0, f.out1
                mdg, mg = (f.out0_h, f.out1_h) if Lp["fn"].rows16 else (mdb, mb)
            else:
                # below the first block only the embedding rows' adjoint in U_dot is left (r_dot^0 = 0)
                ch = ops.RowChain(z.shape[0], False, x.device, fw["x3"])
                rb = ch.stage(P["Wn"], trans=True, in0=hb, res0=rb).out0
                ch.run()
    F, dwf = ops.edge_geom_bwd(d_b, dd_b, d, dd, fw["uhat"], fw["ddel"], topo)
    if not want_theta:
        return fw["U"], F, dwf, None
    uniq, onehot = _species_onehot(z)
    jobs.atb(off(net.atom_embed.weight), onehot, rb, row_map=uniq)
    jobs.run(acc, alpha=-1.0, accumulate=True)                   # w.F = -U_dot
    if accum is not None:
        return fw["U"], F, dwf, None
    return fw["U"], F, dwf, acc.views()


_UNFUSED_MSG = ("this network's shapes are outside the fused interaction-block kernels (n_gaussians > 64, or n_filters above 128 "
                "that is not a multiple of 128 / above 512, or SchNet.fused_block = False): the filter network and the edge-level "
                "products run as library GEMMs (rocBLAS / hipBLASLt) on [E, G] / [E, F] tensors in HBM -- correct, several "
                "times slower than csrc/cfconv_fused.hip")


def _plan_inputs(net, z):
    """(plan, (r0, h0, h0 mirror)) for the one-call evaluation of csrc/schnet_eval.hip, or (None, None)."""
    from . import plan as _plan
    convs = list(net.convolutions)
    Ps = [_layer_params(c) for c in convs]
    fns = [_filter_net(c, P, getattr(c, "filter_bf16", False) or getattr(net, "filter_bf16", False),
                       getattr(net, "node_rows_bf16", False)) for c, P in zip(convs, Ps)]
    pl = _plan.get(net, Ps, fns)
    if pl is None:
        return None, None
    r0, h0 = _first_filter(net, z, Ps[0])
    return pl, (r0, h0, _h0_mirror(net, h0) if fns[0].rows16 else None)


@torch.no_grad()
def force(net, z, x, topo, offsets=None, want_energy=True):
    if fused_ok(net):
        if chain_ok(net):                                         # (no library GEMM on this path: nothing to switch)
            x = x.detach().contiguous()
            pl, rows = _plan_inputs(net, z) if x.is_cuda else (None, None)
            if pl is not None:                                    # the whole evaluation as ONE C-ABI call (nn/plan.py)
                return pl.force(x, topo, rows, want_energy)
            return _force_chain(net, z, x, topo, want_energy)
        with _node_blas():
            return _force_fused(net, z, x.detach().contiguous(), topo, want_energy)
    _warn_once("unfused", _UNFUSED_MSG)
    topo = _stable(topo)
    with _blas_for(topo):
        fw = _primal(net, z, x.detach().contiguous(), topo, topo.offsets)
        return fw["U"], _reverse_U(fw, topo)


@torch.no_grad()
def force_vjp(net, z, x, w, topo, offsets=None, want_theta=True, want_energy=True, accum=None):
    """(U, F, d(w.F)/dx, [d(w.F)/dtheta_p for p in net.parameters()]); the parameter part is skipped
    (None) when want_theta is False, and accumulated into `accum` (ops.ThetaAccum: flat buffer, interval weight read on
    the device) instead of being returned when that is given.  (`offsets` is the topology's own image-flag array; the
    argument is kept for callers that pass it explicitly.)"""
    if fused_ok(net):
        if chain_ok(net):                                         # (no library GEMM on this path: nothing to switch)
            x, w = x.detach().contiguous(), w.detach().contiguous()
            pl, rows = _plan_inputs(net, z) if x.is_cuda else (None, None)
            if pl is not None:                                    # the whole evaluation as ONE C-ABI call (nn/plan.py)
                acc = (accum if accum is not None else ops.ThetaAccum(net.parameters())) if want_theta else None
                U, F, dwf = pl.force_vjp(x, w, topo, rows, z, acc, want_energy)
                return U, F, dwf, (acc.views() if (want_theta and accum is None) else None)
            return _force_vjp_chain(net, z, x, w, topo, want_theta, want_energy, accum)
        with _node_blas():
            return _force_vjp_fused(net, z, x.detach().contiguous(), w.detach().contiguous(), topo, want_theta, want_energy, accum)
    _warn_once("unfused", _UNFUSED_MSG)
    topo = _stable(topo)
    with _blas_for(topo):
        out = _force_vjp_unfused(net, z, x, w, topo, topo.offsets, want_theta)
    if accum is not None and out[3] is not None:
        jobs = ops.GradJobs()
        for p, g in zip(net.parameters(), out[3]):
            jobs.axpy(accum.off[id(p)], g)
        jobs.run(accum, alpha=1.0, accumulate=True)
        return out[0], out[1], out[2], None
    return out


def _force_vjp_unfused(net, z, x, w, topo, offsets, want_theta=True):
    x, w = x.detach().contiguous(), w.detach().contiguous()
    fw = _primal(net, z, x, topo, offsets)
    d, uhat = fw["d"], fw["uhat"]
    # ---------------- tangent sweep along x_dot = w
    ddel = ops._edge_diff(w, topo)
    dd = (uhat * ddel).sum(1)
    rd = None                                                     # r_dot^0 = 0
    for L in fw["layers"]:
        P = L["P"]
        gd = ops.mul_row(L["g"], L["phi"], dd)
        ad = gd.mm(P["W1"].t())
        sa = L["sa"]
        sd = ops.mul_row(sa, ad)
        Wfd = sd.mm(P["W2"].t())
        md = ops._cfconv_agg(L["h"], Wfd, topo)
        hd = None
        if rd is not None:
            hd = rd.mm(P["Wn"].t())
            md = md + ops._cfconv_agg(hd, L["Wf"], topo)
        su = L["su"]
        ud = md.mm(P["U1"].t())
        td = ops.mul_row(su, ud)
        L.update(gd=gd, ad=ad, sa=sa, sd=sd, Wfd=Wfd, hd=hd, md=md, su=su, ud=ud, td=td, rd=rd)
        rd = td.mm(P["U2"].t()) if rd is None else rd + td.mm(P["U2"].t())
    L1, L2, y = fw["L1"], fw["L2"], fw["y"]
    yd = rd.mm(L1.t())
    sy = torch.sigmoid(y)
    # ---------------- reverse sweep of U_dot = sum_i L2 . (sig(y_i) * yd_i)
    ydb = sy * L2
    yb = sy * (1 - sy) * yd * L2
    grads = {}
    ro = net.atomwisereadout.readout["energy"]
    if want_theta:
        grads[id(ro[2].weight)] = (sy * yd).sum(0)[None]
        grads[id(ro[2].bias)] = torch.zeros_like(ro[2].bias)
        grads[id(ro[0].weight)] = yb.t().mm(fw["r"]) + ydb.t().mm(rd)
        grads[id(ro[0].bias)] = yb.sum(0)
    rdb, rb = ydb.mm(L1), yb.mm(L1)
    d_b = torch.zeros_like(d)
    dd_b = torch.zeros_like(d)
    for conv, L in zip(reversed(list(net.convolutions)), reversed(fw["layers"])):
        P = L["P"]
        md_ = conv.moduledict
        tb, tdb = rb.mm(P["U2"]), rdb.mm(P["U2"])
        if want_theta:
            grads[id(md_["update_function"][2].weight)] = rb.t().mm(L["t"]) + rdb.t().mm(L["td"])
            grads[id(md_["update_function"][2].bias)] = rb.sum(0)
        udb, ub = ops.ssp_dual_bwd(L["su"], L["ud"], tdb, tb)
        mdb, mb = udb.mm(P["U1"]), ub.mm(P["U1"])
        if want_theta:
            grads[id(md_["update_function"][0].weight)] = udb.t().mm(L["md"]) + ub.t().mm(L["m"])
            grads[id(md_["update_function"][0].bias)] = ub.sum(0)
        hdb = ops._cfconv_agg(mdb, L["Wf"], topo)
        hb = ops._cfconv_agg(mdb, L["Wfd"], topo) + ops._cfconv_agg(mb, L["Wf"], topo)
        Wfb = ops._edge_prod(mb, L["h"], topo)
        if L["hd"] is not None:
            Wfb = Wfb + ops._edge_prod(mdb, L["hd"], topo)
        Wfdb = ops._edge_prod(mdb, L["h"], topo)
        if want_theta:
            gWn = hb.t().mm(L["r"])
            if L["rd"] is not None:
                gWn = gWn + hdb.t().mm(L["rd"])
            grads[id(md_["message_node_filter"].weight)] = gWn
            grads[id(md_["message_node_filter"].bias)] = hb.sum(0)
        rdb = rdb + hdb.mm(P["Wn"])
        rb = rb + hb.mm(P["Wn"])
        # filter network
        E_ = Wfb.shape[0]
        if E_ <= _CAT_EDGES:
            # launch-bound sizes: primal + tangent adjoints stacked so that each pair of GEMMs is one
            cat_w = torch.cat((Wfdb, Wfb))
            both = cat_w.mm(P["W2"])
            sdb, sb = both[:E_], both[E_:]
            adb, ab = ops.ssp_dual_bwd(L["sa"], L["ad"], sdb, sb)
            both = torch.cat((adb, ab))
            bg = both.mm(P["W1"])
            gdb, gb = bg[:E_], bg[E_:]
            if want_theta:
                grads[id(md_["message_edge_filter"][3].weight)] = _atb(cat_w, torch.cat((L["sd"], L["s"])))
                grads[id(md_["message_edge_filter"][1].weight)] = _atb(both, torch.cat((L["gd"], L["g"])))
        else:
            # bandwidth-bound sizes: the [2E, F] copies would cost more than the launches they save
            sdb, sb = Wfdb.mm(P["W2"]), Wfb.mm(P["W2"])
            adb, ab = ops.ssp_dual_bwd(L["sa"], L["ad"], sdb, sb)
            gdb, gb = adb.mm(P["W1"]), ab.mm(P["W1"])
            if want_theta:
                grads[id(md_["message_edge_filter"][3].weight)] = _atb(Wfdb, L["sd"]) + _atb(Wfb, L["s"])
                grads[id(md_["message_edge_filter"][1].weight)] = _atb(adb, L["gd"]) + _atb(ab, L["g"])
        if want_theta:
            grads[id(md_["message_edge_filter"][3].bias)] = Wfb.sum(0)
            grads[id(md_["message_edge_filter"][1].bias)] = ab.sum(0)
        ops.smear_bwd(gdb, gb, L["g"], L["phi"], dd, P["c"], d_b, dd_b)
    # geometry: dd = uhat . ddel, d = |delta|;  dd_b = dU/dd (module docstring), so the force comes from this sweep too
    delta_b = d_b[:, None] * uhat + (dd_b / d)[:, None] * (ddel - dd[:, None] * uhat)
    xb = ops._edge_scatter(delta_b, topo)
    F = -ops._edge_scatter(dd_b[:, None] * uhat, topo)
    if not want_theta:
        return fw["U"], F, -xb, None
    # embedding rows: one-hot(z)^T rb as a GEMM (no float atomics: index_add_ on a handful of species
    # serialises and is not reproducible)
    uniq, onehot = _species_onehot(z)
    emb = torch.zeros_like(net.atom_embed.weight)
    emb[uniq] = _atb(onehot, rb)                  # [S, N] x [N, A] on the split-K kernel when N is large
    grads[id(net.atom_embed.weight)] = emb
    # w.F = -U_dot
    # one negation for all parameters (views of a flat buffer) instead of one tiny kernel per tensor
    plist = list(net.parameters())
    flat = torch.cat([grads[id(p)].reshape(-1) for p in plist]).neg_()
    out, pos = [], 0
    for p in plist:
        out.append(flat[pos:pos + p.numel()].reshape(p.shape))
        pos += p.numel()
    return fw["U"], F, -xb, out
