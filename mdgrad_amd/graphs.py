"""HIP-graph replay of the generic (non-fused) NoseHooverChain integrator steps.

A SchNet force + Hessian-vector evaluation is ~300 kernel launches of a few microseconds each: below
~10^5 edges the host cannot issue them as fast as the GPU retires them (measured: GPU busy 7-27 %).  One
forward NH-Verlet step, and one adjoint interval (torchmd/sovlers.py:129-164 / 253-288), are therefore
captured once into a HIP graph and replayed per step.  What makes the capture possible:

  * fixed-capacity neighbour lists (interface.*.set_static_topology): the rebuild inside the step has no
    host-side pair count; padding rows are inert (ops.StaticTopo); an overflow is detected after the
    pass (one host sync), the capacities grow, and the pass is redone eagerly;
  * the frame index lives on the device and is advanced by the graph itself, the saved trajectory and
    the incoming frame gradients are copied once per pass into static buffers, so a pass of T-1 steps is
    T-1 graph launches and nothing else.

The arithmetic is the eager path's (same functions, same order): results are bitwise identical.
Graphs are cached on the integrator per (kind, shapes, frames, capacities, parameter identities).
"""
import gc
import os

import torch

from .tinydiffeq import _flatten

# Largest fixed edge capacity that is still replayed from a captured graph; larger systems run eagerly on the same
# fixed-capacity lists (eager_static: no host sync per rebuild).  Replay was measured up to 2 M edges: at 4096 beads x 8
# replicas it is no faster than the sync-free eager pass (the GPU is busy 75 of 85 ms either way), and capturing
# half-million-pair graphs from the autograd thread aborted once in three runs on ROCm 7.2, so the limit stays where
# launches, not kernels, bound the step.
MAX_EDGES = int(os.environ.get("MDG_GRAPH_MAX_EDGES", str(1 << 18)))


def enabled(func):
    """Graph replay applies to an integrator with the analytic-adjoint protocol, a neighbour rebuild
    at every call and fixed-capacity lists available on every member of its model."""
    if os.environ.get("MDG_GRAPHS", "1") == "0" or not torch.cuda.is_available():
        return False
    if getattr(func, "use_graphs", True) is False:
        return False
    model = getattr(func, "model", None)
    if model is None or getattr(func, "topology_update_freq", 0) != 1 or not hasattr(func, "rhs_vjp"):
        return False
    if not getattr(func, "supports_rhs_vjp", lambda: False)():
        return False
    if not getattr(model, "supports_static_topology", lambda: False)():
        return False
    return func.mass.is_cuda


def _capture(body, reset):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):                       # warm-up: library handles, caches, allocator pool
            reset()
            body()
    torch.cuda.current_stream().wait_stream(side)
    reset()
    # No cyclic garbage collection while the stream is capturing: an integrator that went out of scope earlier keeps
    # its graphs alive through a reference cycle (integrator -> cache -> graph -> integrator), and the collector
    # destroying such a graph (hipGraphExecDestroy + its private pool) in the middle of a capture aborts the process
    # on ROCm 7.2 (seen once in three cold runs).  Collect first, then hold the collector off until the capture ends.
    was_enabled = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
    finally:
        if was_enabled:
            gc.enable()
    return g


def _too_large(func):
    for m in getattr(func.model, "models", {"m": func.model}).values():
        st = getattr(m, "_static", None)
        if st is not None and st.get("capacity", 0) > MAX_EDGES:
            return True
    return False


def _key(func, kind, y, T):
    # (the thermostat temperature is read from a device scalar, see NoseHooverChain._T_device: annealing
    #  schedules that call update_T every epoch keep their graphs)
    return (kind, tuple(tuple(x.shape) for x in y), int(T), func.model.static_version(),
            tuple(id(p) for p in func.parameters()))


class _ForwardGraph:
    """v, q, pv, F <- one NH-Verlet step (NHVerlet.integrate's loop body); frame k+1 stored."""

    def __init__(self, func, y0, T):
        dev = y0[0].device
        self.state = [x.clone() for x in y0]
        self.F = torch.zeros_like(y0[1])
        self.out = [torch.zeros((T,) + tuple(x.shape), device=dev, dtype=x.dtype) for x in y0]
        self.t = torch.zeros(T, device=dev, dtype=y0[0].dtype)
        self.k = torch.zeros(1, dtype=torch.int64, device=dev)
        self.func = func
        self.graph = _capture(self._step, self.k.zero_)

    def _step(self):
        if len(self.state) == 2:                  # NVE (velocity Verlet, sovlers.py:25-33; dv/dt = F)
            func, (v, q), F, k = self.func, self.state, self.F, self.k
            dt = self.t.index_select(0, k + 1) - self.t.index_select(0, k)
            dv_h = 0.5 * F * dt
            dq = (v + dv_h) * dt
            qn = q + dq
            Fn = func.force(qn)
            vn = v + (dv_h + 0.5 * Fn * dt)
            v.copy_(vn), q.copy_(qn), F.copy_(Fn)
            for o, x in zip(self.out, (v, q)):
                o.index_copy_(0, k + 1, x[None])
            k.add_(1)
            return
        func, (v, q, pv), F, k = self.func, self.state, self.F, self.k
        if getattr(func, "fused_steps_ok", lambda *a: False)(v, q, pv):
            # both halves of the step as one launch each (csrc/nhc.hip): rhs + half kick + drift, force, rhs + finish
            # + frame store
            w = func.nhv_work(v, pv)
            qn = w.kick(v, q, pv, F, self.t, k)
            w.finish(v, q, pv, F, func.force(qn), self.t, k, self.out, advance=True)    # (k <- k + 1 inside)
            return
        dt = self.t.index_select(0, k + 1) - self.t.index_select(0, k)
        a0, _, b0 = func.rhs_from_force((v, q, pv), F)
        dv_h = 1 / 2 * a0 * dt
        dp_h = 1 / 2 * b0 * dt
        dq = (v + dv_h) * dt
        qn = q + dq
        Fn = func.force(qn)
        a1, _, b1 = func.rhs_from_force((v + dv_h, qn, pv + dp_h), Fn)
        vn, pn = v + (dv_h + 1 / 2 * a1 * dt), pv + (dp_h + 1 / 2 * b1 * dt)
        v.copy_(vn), q.copy_(qn), pv.copy_(pn), F.copy_(Fn)
        for o, x in zip(self.out, (v, q, pv)):
            o.index_copy_(0, k + 1, x[None])
        k.add_(1)

    def run(self, y0, t):
        func = self.func
        getattr(func.model, "prepare_pass", lambda: None)()
        for s, x in zip(self.state, y0):
            s.copy_(x)
        self.t.copy_(t)
        for o, x in zip(self.out, y0):
            o[0].copy_(x)
        c0 = func.update_count
        self.F.copy_(func.force(self.state[1]))
        self.k.zero_()
        for _ in range(t.shape[0] - 1):
            self.graph.replay()
        func.update_count = c0 + 2 * (t.shape[0] - 1)      # the reference's two calls per step (sovlers.py:111,121)
        return tuple(o.clone() for o in self.out)


class _AdjointGraph:
    """lam, gth <- one interval of _analytic_nhc_adjoint; the frame index counts down on the device."""

    def __init__(self, func, ans, n_params):
        dev = ans[0].device
        self.func = func
        self.ans = [torch.zeros_like(a) for a in ans]
        self.gout = [torch.zeros_like(a) for a in ans]
        self.t = torch.zeros(ans[0].shape[0], device=dev, dtype=ans[0].dtype)
        self.lam = [torch.zeros_like(a[0]) for a in ans]
        self.gth = torch.zeros(n_params, device=dev, dtype=ans[0].dtype)
        self.i = torch.ones(1, dtype=torch.int64, device=dev)
        for a, src in zip(self.ans, ans):          # a physical state for the warm-up / capture evaluations
            a.copy_(src)
        self.graph = _capture(self._interval, lambda: self.i.fill_(1))

    def _interval(self):
        func, lam, i = self.func, self.lam, self.i
        if len(lam) == 2:                         # NVE: sovlers.nve_adjoint_interval on the frame the device index names
            from .sovlers import nve_adjoint_interval
            h = self.t.index_select(0, i) - self.t.index_select(0, i - 1)
            v, x = (a.index_select(0, i)[0] for a in self.ans)
            g_prev = tuple(g.index_select(0, i - 1)[0] for g in self.gout)
            new, th0 = nve_adjoint_interval(func, v, x, (lam[0], lam[1]), h, g_prev)
            lam[0].copy_(new[0]), lam[1].copy_(new[1])
            if th0:
                self.gth.add_(_flatten(th0) * 0.5 * h * 2)
            i.sub_(1)
            return
        if getattr(func, "fused_steps_ok", lambda *a: False)(*lam):
            # sovlers.py:258 (counter / rebuild only), two force-vjp evaluations, three launches of algebra around them
            w = func.nhv_work(lam[0], lam[2])
            q, wv = w.adj_pre(self.ans, lam[0], i)
            func.update_topology(q)
            func.update_topology(q)
            F, dwf, _ = func.model.force_vjp(q, wv, want_theta=False)
            qm, wh = w.adj_mid(lam, F, dwf, self.t, i)
            func.update_topology(qm)
            if getattr(func.model, "accepts_accum", False):
                # every piece of the parameter gradient is added into self.gth by the kernel that reduces it, weighted
                # with t[i] - t[i-1] read on the device (sovlers.py:160)
                from . import ops
                acc = ops.ThetaAccum(func.parameters(), flat=self.gth, t=self.t, idx=i)
                _, dwf1, th1 = func.model.force_vjp(qm, wh, accum=acc)
            else:
                _, dwf1, th1 = func.model.force_vjp(qm, wh)
            w.adj_end(lam, dwf1, self.t, i, self.gout, advance=not th1)                 # (i <- i - 1 inside)
            if th1:
                self.gth.add_(_flatten(func.theta_in_parameter_order(th1)) * (self.t.index_select(0, i) - self.t.index_select(0, i - 1)))
                i.sub_(1)
            return
        h = self.t.index_select(0, i) - self.t.index_select(0, i - 1)
        v, q, pv = (a.index_select(0, i)[0] for a in self.ans)
        func.update_topology(q)                                   # sovlers.py:258 (counter / rebuild only)
        (a, _, b), G0, _ = func.rhs_vjp((v, q, pv), lam, want_theta=False)
        hh = 0.5 * h
        vh = v - a * hh                                           # :132
        qm = q + vh * h                                           # :138 (forward-time sign)
        pm = pv - b * hh                                          # :135
        lam_h = [l + g * hh for l, g in zip(lam, G0)]             # :141-143
        _, G1, th1 = func.rhs_vjp((vh, qm, pm), lam_h)
        for k in range(3):
            lam[k].copy_(lam[k] + G1[k] * h + self.gout[k].index_select(0, i - 1)[0])   # :156-158, :286
        if th1:
            self.gth.add_(_flatten(th1) * h)                      # :160
        i.sub_(1)

    def run(self, t, ans, grad_output):
        T = ans[0].shape[0]
        getattr(self.func.model, "prepare_pass", lambda: None)()
        for dst, src in zip(self.ans, ans):
            dst.copy_(src)
        for dst, src in zip(self.gout, grad_output):
            dst.copy_(src)
        self.t.copy_(t)
        for l, g in zip(self.lam, grad_output):
            l.copy_(g[-1])
        self.gth.zero_()
        self.i.fill_(T - 1)
        for _ in range(T - 1):
            self.graph.replay()
        self.func.update_count += 3 * (T - 1)
        return [l.clone() for l in self.lam], self.gth.clone()


def _cache(func):
    c = getattr(func, "_graph_cache", None)
    if c is None:
        c = func._graph_cache = {}
    return c


def _prepare(func):
    """Switch the model to fixed-capacity lists (sized from its current topology the first time)."""
    func.model.set_static_topology(True)
    return not _too_large(func)


def _finish(func, cache, q_last):
    """Back to exact-size lists for whatever runs next (rebuilt at q_last: the padded list must not leak
    to the autograd path or to user code reading nbr_list); True when this pass overflowed its capacities
    (they have been enlarged, the graphs captured with the old ones are dropped)."""
    overflow = func.model.static_overflow()
    func.model.set_static_topology(False)
    func.model._reset_topology(q_last)
    if overflow:
        cache.clear()
    return overflow


def eager_static(func, body, q_last_of):
    """Run `body()` (an eager integration / adjoint sweep) on fixed-capacity neighbour lists: systems too large for
    graph replay still profit from them, because a rebuild then needs NO host sync (the exact-size path reads the
    longest row and the pair count back at every rebuild, which keeps the launch thread from running ahead of the
    GPU).  Capacities are checked once after the pass; on overflow they grow and None is returned (the caller
    redoes the pass on exact-size lists)."""
    if not enabled(func):
        return None
    func.model.set_static_topology(True)
    try:
        out = body()
    finally:
        overflow = func.model.static_overflow()
        func.model.set_static_topology(False)
    func.model._reset_topology(q_last_of(out))
    return None if overflow else out


def forward(func, y0, t):
    """Frames (v_t, q_t, pv_t) of NHVerlet.integrate by graph replay, or None when graphs do not apply or
    the capacities overflowed (the caller then integrates eagerly)."""
    cache = _cache(func)
    if not _prepare(func):
        func.model.set_static_topology(False)
        return None
    key = _key(func, "fwd", y0, t.shape[0])
    g = cache.get(key)
    if g is None:
        func.model._reset_topology(y0[1])
        for k_ in [k_ for k_ in cache if k_[0] == "fwd"]:
            del cache[k_]
        g = cache[key] = _ForwardGraph(func, y0, t.shape[0])
    out = g.run(y0, t)
    return None if _finish(func, cache, out[1][-1]) else out


def adjoint(func, t, ans, grad_output, n_params):
    """(lam, gth) of _analytic_nhc_adjoint by graph replay, or None (see forward)."""
    cache = _cache(func)
    if not _prepare(func):
        func.model.set_static_topology(False)
        return None
    key = _key(func, "adj", ans, ans[0].shape[0])
    g = cache.get(key)
    if g is None:
        func.model._reset_topology(ans[1][0])
        for k_ in [k_ for k_ in cache if k_[0] == "adj"]:
            del cache[k_]
        g = cache[key] = _AdjointGraph(func, ans, n_params)
    out = g.run(t, ans, grad_output)
    return None if _finish(func, cache, ans[1][0]) else out
