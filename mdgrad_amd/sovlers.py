"""Solvers and the adjoint engine (module name keeps the reference's spelling,
torchmd/sovlers.py; `mdgrad_amd.solvers` is an alias).

Two execution paths behind the same odeint / odeint_adjoint API:

  fused    the integrator is a NoseHooverChain / NVE over built-in pair potentials with
           topology_update_freq == 1 and the matching method ('NH_verlet' / 'verlet'):
           ops.FusedTrajFn runs the whole forward trajectory in ONE HIP launch and the whole
           adjoint sweep in ONE launch (csrc/traj_small.hip).
  generic  anything else (GNN potentials, user modules, stale neighbour lists, rk4): the
           reference's Python control flow, with every energy evaluation and its first and
           second derivatives served by HIP autograd ops.  Call pattern, update order and
           the topology counter are the reference's (sovlers.py:106-168, 196-293).
"""
import torch
from torch import nn

from . import graphs, ops
from .tinydiffeq import (FixedGridODESolver, RK4, _check_inputs, _flatten,
                         _flatten_convert_none_to_zeros)


def NHverlet_update(func, t, dt, y):
    """One NH-Verlet step as increments (sovlers.py:106-168).  3 states = forward, 8 states =
    augmented (state, adjoint, adj_time, adj_params) integrated on the reversed grid."""
    if len(y) == 3:
        v, q, pv = y
        a0, _, b0 = func(t, y)
        dv_h = 1 / 2 * a0 * dt
        dp_h = 1 / 2 * b0 * dt
        dq = (v + dv_h) * dt
        a1, _, b1 = func(t, (v + dv_h, q + dq, pv + dp_h))
        return (dv_h + 1 / 2 * a1 * dt, dq, dp_h + 1 / 2 * b1 * dt)
    if len(y) == 8:
        k0 = func(t, y)
        dv_h = 1 / 2 * k0[0] * dt
        dp_h = 1 / 2 * k0[2] * dt
        dq = (y[0] + dv_h) * dt                       # forward-time sign, as the reference
        half = tuple(k0[i] * 0.5 * dt for i in range(3, 8))
        k1 = func(t, (y[0] + dv_h, y[1] + dq, y[2] + dp_h) +
                  tuple(y[i] + half[i - 3] for i in range(3, 8)))
        return (dv_h + 1 / 2 * k1[0] * dt, dq, dp_h + 1 / 2 * k1[2] * dt) + \
            tuple(k1[i] * dt for i in range(3, 8))
    raise ValueError("NHverlet_update takes 3 state tensors (forward) or 8 (augmented adjoint state), got %d" % len(y))


def verlet_update(func, t, dt, y):
    """One velocity-Verlet step as increments (sovlers.py:21-104); 2 states forward, 6 augmented."""
    if len(y) == 2:
        v, q = y
        a0, _ = func(t, y)
        dv_h = 0.5 * a0 * dt
        dq = (v + dv_h) * dt
        a1, _ = func(t, (v + dv_h, q + dq))
        return (dv_h + 0.5 * a1 * dt, dq)
    if len(y) == 6:
        v, x, lv, lx = y[0], y[1], y[2], y[3]
        dv, _, _, X0, vjp_t, Th0 = func(t, y)
        dv_h = 1 / 2 * dv * dt
        v_half = v - dv_h
        dx = v_half * dt
        x0 = x - dx
        dlx = X0 * dt * 0.5
        dlv = (lx + dlx) * dt
        dth_half = Th0 * 0.5 * dt
        dv2, _, _, X1, vjp_t2, _ = func(t, (v_half, x0, lv + dlv, lx + dlx, y[4] + vjp_t * dt,
                                            y[5] + dth_half))
        return (dv_h - dv2 * dt * 0.5, dx, dlv, X1 * dt * 0.5 + dlx, vjp_t2 * dt, dth_half * 2)
    raise ValueError("verlet_update takes 2 state tensors (forward) or 6 (augmented adjoint state), got %d" % len(y))


class NHVerlet(FixedGridODESolver):
    def step_func(self, func, t, dt, y):
        return NHverlet_update(func, t, dt, y)

    def integrate(self, t):
        """Graph-free forward integration of a NoseHooverChain with the force at q_{k+1} evaluated once
        and reused by the next step (the reference evaluates it twice, sovlers.py:111,121 -- same q,
        same value).  Anything else goes through the reference's two-call step."""
        func = self.func
        if (torch.is_grad_enabled() or len(self.y0) != 3 or not hasattr(func, "rhs_from_force")
                or getattr(func, "topology_update_freq", 0) != 1):
            return super().integrate(t)
        t = t.type_as(self.y0[0]).to(self.y0[0].device)

        def eager():
            out = eager_()
            # the call counter as the reference leaves it: two right-hand-side calls per step (sovlers.py:111,121), whatever
            # the number of force evaluations made here (the counter decides rebuilds once the frequency changes)
            func.update_count = c0 + 2 * (t.shape[0] - 1)
            return out

        c0 = getattr(func, "update_count", 0)

        def eager_():
            v, q, pv = self.y0
            if getattr(func, "fused_steps_ok", lambda *a: False)(v, q, pv):
                return _nhv_forward(func, self.y0, t)
            frames = [(v, q, pv)]
            F = func.force(q)
            for k in range(t.shape[0] - 1):
                dt = t[k + 1] - t[k]
                a0, _, b0 = func.rhs_from_force((v, q, pv), F)
                dv_h = 1 / 2 * a0 * dt
                dp_h = 1 / 2 * b0 * dt
                dq = (v + dv_h) * dt
                F = func.force(q + dq)
                a1, _, b1 = func.rhs_from_force((v + dv_h, q + dq, pv + dp_h), F)
                v, q, pv = v + (dv_h + 1 / 2 * a1 * dt), q + dq, pv + (dp_h + 1 / 2 * b1 * dt)
                frames.append((v, q, pv))
            return tuple(torch.stack([f[i] for f in frames]) for i in range(3))

        if graphs.enabled(func) and t.shape[0] > 3:
            out = graphs.forward(func, tuple(self.y0), t)          # HIP-graph replay, one launch per step
            if out is None:                                         # too large for replay: eager, but sync-free lists
                out = graphs.eager_static(func, eager, lambda o: o[1][-1])
            if out is not None:
                return out
        return eager()


def _nhv_forward(func, y0, t):
    """The loop of NHVerlet.integrate with each half of a step as ONE launch (csrc/nhc.hip mdg_nhv_kick /
    mdg_nhv_finish: rhs + half kick + drift; rhs + finish + frame store) around the force evaluation."""
    v, q, pv = (x.clone() for x in y0)
    T = t.shape[0]
    out = [torch.empty((T,) + tuple(x.shape), device=x.device, dtype=x.dtype) for x in y0]
    for o, x in zip(out, y0):
        o[0].copy_(x)
    F = func.force(q).contiguous().clone()
    w = func.nhv_work(v, pv)
    k = torch.zeros(1, dtype=torch.int64, device=v.device)
    t = t.contiguous()
    for _ in range(T - 1):
        qn = w.kick(v, q, pv, F, t, k)
        w.finish(v, q, pv, F, func.force(qn), t, k, out, advance=True)          # (k <- k + 1 inside the launch)
    return tuple(out)


class Verlet(FixedGridODESolver):
    def step_func(self, func, t, dt, y):
        return verlet_update(func, t, dt, y)

    def integrate(self, t):
        """Graph-free forward integration of an NVE integrator whose model has analytic forces: the force at q_{k+1} is
        evaluated once and reused by the next step (the reference's two calls per step, sovlers.py:25-33, see the same q),
        and the steps are replayed from a HIP graph where that applies -- NHVerlet.integrate for the 2-state system."""
        func = self.func
        if (torch.is_grad_enabled() or len(self.y0) != 2 or not hasattr(func, "force")
                or not getattr(func, "supports_rhs_vjp", lambda: False)()
                or getattr(func, "topology_update_freq", 0) != 1 or getattr(func, "analytic_verlet", True) is False):
            return super().integrate(t)
        t = t.type_as(self.y0[0]).to(self.y0[0].device)

        def eager():
            out = eager_()
            func.update_count = c0 + 2 * (t.shape[0] - 1)          # (as NHVerlet.integrate: the reference's count)
            return out

        c0 = getattr(func, "update_count", 0)

        def eager_():
            v, q = self.y0
            frames = [(v, q)]
            F = func.force(q)                                     # (NVE: dv/dt = F, no 1/m -- md.py:145-148)
            for k in range(t.shape[0] - 1):
                dt = t[k + 1] - t[k]
                dv_h = 0.5 * F * dt
                dq = (v + dv_h) * dt
                F = func.force(q + dq)
                v, q = v + (dv_h + 0.5 * F * dt), q + dq
                frames.append((v, q))
            return tuple(torch.stack([f[i] for f in frames]) for i in range(2))

        if graphs.enabled(func) and t.shape[0] > 3:
            out = graphs.forward(func, tuple(self.y0), t)
            if out is None:
                out = graphs.eager_static(func, eager, lambda o: o[1][-1])
            if out is not None:
                return out
        return eager()


SOLVERS = {'rk4': RK4, 'NH_verlet': NHVerlet, 'verlet': Verlet}


def odeint(func, y0, t, rtol=1e-7, atol=1e-9, method=None, options=None):
    """sovlers.py:171-193 (generic path)."""
    tensor_input, func, y0, t = _check_inputs(func, y0, t)
    if options is None:
        options = {}
    elif method is None:
        raise ValueError('cannot supply `options` without specifying `method`')
    if method not in SOLVERS:
        raise KeyError(method)
    solution = SOLVERS[method](func, y0, rtol=rtol, atol=atol, **options).integrate(t)
    return solution[0] if tensor_input else solution


class _BackwardSystem:
    """Right-hand side of the system the generic adjoint integrates backwards over one interval
    (torchmd/sovlers.py:221-245): packed state = (y_1..y_n, lam_1..lam_n, accumulated dL/dt, accumulated
    dL/dtheta).  d(y)/dt = f(t, y); the other three blocks are the vector-Jacobian products of f with -lam, taken
    either in closed form (integrators that implement `rhs_vjp`) or by autograd through `func`."""

    def __init__(self, func, n_state, closed_form):
        self.func, self.n, self.closed_form = func, n_state, closed_form
        self.theta = tuple(func.parameters())

    def __call__(self, time, packed):
        state, costate = packed[:self.n], packed[self.n:2 * self.n]
        if self.closed_form:
            # rhs_vjp returns lam^T df/d(y, theta); the reference contracts with -lam (:232), hence the signs
            rhs, by_state, by_theta = self.func.rhs_vjp(tuple(x.detach() for x in state), tuple(x.detach() for x in costate))
            by_theta = _flatten([-g for g in by_theta]) if len(by_theta) else torch.tensor(0.).to(state[0])
            return (*rhs, *(-g for g in by_state), torch.zeros(()).to(state[0]), by_theta)
        with torch.enable_grad():
            time = time.to(state[0].device).detach().requires_grad_(True)
            state = tuple(x.detach().requires_grad_(True) for x in state)
            rhs = self.func(time, state)
            by_time, *others = torch.autograd.grad(rhs, (time,) + state + self.theta, tuple(-c for c in costate),
                                                   allow_unused=True, retain_graph=True)
        by_state = tuple(torch.zeros_like(x) if g is None else g for g, x in zip(others[:self.n], state))
        by_theta = (_flatten_convert_none_to_zeros(others[self.n:], self.theta) if self.theta
                    else torch.tensor(0.).to(by_state[0]))
        return (*rhs, *by_state, torch.zeros_like(time) if by_time is None else by_time, by_theta)


class OdeintAdjointMethod(torch.autograd.Function):
    """Generic adjoint (torchmd/sovlers.py:196-293) for whatever the fused / analytic paths do not take: the forward
    pass keeps no graph; the backward pass walks the saved frames from the last to the first, integrating
    `_BackwardSystem` over each interval with the SAME solver (so `NH_verlet` / `verlet` run their backward branches,
    :129-164 / :42-101) and adding the incoming frame cotangents at every grid point (:286)."""

    @staticmethod
    def forward(ctx, *args):
        *y0, func, t, flat_params, rtol, atol, method, options = args
        ctx.func, ctx.solver = func, dict(rtol=rtol, atol=atol, method=method, options=options)
        with torch.no_grad():
            frames = odeint(func, tuple(y0), t, **ctx.solver)
        ctx.save_for_backward(t, flat_params, *frames)
        return frames

    @staticmethod
    def backward(ctx, *cotangents):
        t, flat_params, *saved = ctx.saved_tensors
        frames, n = tuple(saved), len(saved)
        func, solver = ctx.func, ctx.solver
        wants_time = ctx.needs_input_grad[n + 1]
        closed_form = (not wants_time) and getattr(func, "supports_rhs_vjp", lambda: False)()
        if closed_form and solver["method"] == 'NH_verlet' and n == 3:
            return _analytic_nhc_adjoint(func, t, frames, cotangents, flat_params)
        if closed_form and solver["method"] == 'verlet' and n == 2 and getattr(func, "analytic_verlet", True) is not False:
            return _analytic_nve_adjoint(func, t, frames, cotangents, flat_params)
        system = _BackwardSystem(func, n, closed_form)
        with torch.no_grad():
            costate = tuple(c[-1] for c in cotangents)                     # lam(t_last) = dL/dy_last   (:249)
            theta_bar = torch.zeros_like(flat_params) if flat_params.numel() else torch.tensor(0.).to(costate[0])
            time_bar = torch.tensor(0.).to(t)
            per_point = []                                                 # dL/dt_i, last grid point first
            for i in reversed(range(1, frames[0].shape[0])):
                y_i = tuple(x[i] for x in frames)
                if wants_time or not hasattr(func, "update_topology"):
                    f_i = func(t[i], y_i)                                  # :258
                    sens = sum(torch.dot(f.reshape(-1), c[i].reshape(-1)).reshape(1) for f, c in zip(f_i, cotangents))
                else:
                    # the reference evaluates func here only for dL/dt, which nobody consumes when t needs no
                    # gradient; keep its side effect (the topology counter / rebuild, md.py:200-204), skip the force
                    func.update_topology(y_i[1])
                    sens = torch.zeros(1).to(t)
                time_bar = time_bar - sens
                per_point.append(sens)
                packed = odeint(system, (*y_i, *costate, time_bar, theta_bar), torch.stack([t[i], t[i - 1]]), **solver)
                costate = tuple(lam[1] + c[i - 1] for lam, c in zip(packed[n:2 * n], cotangents))
                time_bar, theta_bar = packed[2 * n][1], packed[2 * n + 1][1]
            per_point.append(time_bar)
            time_grad = torch.cat([x.reshape(-1) for x in reversed(per_point)])
        return (*costate, None, time_grad, theta_bar, None, None, None, None, None)


def _analytic_nhc_adjoint(func, t, ans, grad_output, flat_params):
    """The same sweep as OdeintAdjointMethod.backward + the backward branch of NHverlet_update
    (sovlers.py:129-164, 253-288) for integrators that provide rhs_vjp: per interval the counter-only
    call, two analytic augmented evaluations and the explicit-midpoint adjoint update, written with
    the handful of tensor ops it needs instead of 8-tuple solver algebra."""
    with torch.no_grad():
        T = ans[0].shape[0]

        def eager():
            lam = [g[-1].clone() for g in grad_output]
            gth = torch.zeros_like(flat_params)
            if getattr(func, "fused_steps_ok", lambda *a: False)(*lam):
                # per interval: frame gather + w = lam_v / m, force-vjp, midpoint algebra, force-vjp, adjoint update --
                # three launches of algebra (csrc/nhc.hip mdg_nhv_adj_*) instead of ~45 tensor ops
                w = func.nhv_work(lam[0], lam[2])
                frames = [a.contiguous() for a in ans]
                gout = [g.contiguous() for g in grad_output]
                # the mdg_nhv_adj_* kernels read the time grid as a float32 device array (a float64 or host grid handed
                # straight to odeint_adjoint must not reach them as a raw pointer)
                tc = t.to(device=lam[0].device, dtype=lam[0].dtype).contiguous()
                idx = torch.full((1,), T - 1, dtype=torch.int64, device=lam[0].device)
                # parameter gradients: every kernel that produces a piece adds  (t[i] - t[i-1]) * piece  straight into the
                # flat buffer (:160), reading the interval from the device grid
                accepts = getattr(func.model, "accepts_accum", False)
                acc = ops.ThetaAccum(func.parameters(), flat=gth, t=tc, idx=idx) if accepts else None
                for i in range(T - 1, 0, -1):
                    q, wv = w.adj_pre(frames, lam[0], idx)
                    func.update_topology(q)                               # :258 (dL/dt call: counter / rebuild only)
                    func.update_topology(q)
                    F, dwf, _ = func.model.force_vjp(q, wv, want_theta=False)
                    qm, wh = w.adj_mid(lam, F, dwf, tc, idx)
                    func.update_topology(qm)
                    if accepts:
                        _, dwf1, th1 = func.model.force_vjp(qm, wh, accum=acc)
                    else:
                        _, dwf1, th1 = func.model.force_vjp(qm, wh)
                    w.adj_end(lam, dwf1, tc, idx, gout, advance=True)             # (idx <- idx - 1 inside the launch)
                    if th1:
                        gth += _flatten(func.theta_in_parameter_order(th1)) * (tc[i] - tc[i - 1])   # :160
                return lam, gth
            for i in range(T - 1, 0, -1):
                h = t[i] - t[i - 1]
                v, q, pv = ans[0][i], ans[1][i], ans[2][i]
                func.update_topology(q)                                   # :258 (dL/dt call: counter / rebuild only)
                (a, _, b), G0, _ = func.rhs_vjp((v, q, pv), lam, want_theta=False)
                hh = 0.5 * h
                vh = v - a * hh                                           # :132  v + 1/2 (-a) h
                qm = q + vh * h                                           # :138  forward-time sign (quirk)
                pm = pv - b * hh                                          # :135
                lam_h = [l + g * hh for l, g in zip(lam, G0)]             # :141-143
                _, G1, th1 = func.rhs_vjp((vh, qm, pm), lam_h)
                for k in range(3):
                    lam[k] = lam[k] + G1[k] * h + grad_output[k][i - 1]   # :156-158, :286
                if th1:
                    gth = gth + _flatten(th1) * h                         # :160
            return lam, gth

        out = None
        if graphs.enabled(func) and T > 3:
            out = graphs.adjoint(func, t, ans, grad_output, flat_params.numel())   # one graph launch per interval
            if out is None:                                               # too large for replay: sync-free lists
                out = graphs.eager_static(func, eager, lambda o: ans[1][0])
        if out is None:
            out = eager()
        return (*out[0], None, None, out[1], None, None, None, None, None)


def nve_adjoint_interval(func, v, x, lam, h, g_prev):
    """One interval of OdeintAdjointMethod.backward over the backward branch of verlet_update (sovlers.py:42-101, :253-288)
    for an integrator with rhs_vjp: (new costate, parameter term).  h = t[i] - t[i-1] > 0: the generic path integrates the
    interval in s = -t with the NEGATED augmented right-hand side (tinydiffeq.py:132-135), i.e. the solver sees
    (-F, -v, +lam_q, +d(lam_v.F)/dq, 0, +d(lam_v.F)/dtheta) and reads the first, fourth and sixth entry of it, twice: at
    (x, lam) with the parameter term and at the half-updated point without."""
    lv, lx = lam
    func.update_topology(x)                                       # :258 (the dL/dt call: counter / rebuild only)
    (F0, _), (_, X0), th0 = func.rhs_vjp((v, x), (lv, lx))
    dv_h = 1 / 2 * (-F0) * h
    v_half = v - dv_h                                             # :49-50
    dx = v_half * h
    x0 = x - dx                                                   # :51-52
    dlx = X0 * h * 0.5                                            # :71
    dlv = (lx + dlx) * h                                          # :72
    _, (_, X1), _ = func.rhs_vjp((v_half, x0), (lv + dlv, lx + dlx), want_theta=False)
    lam_new = (lv + dlv + g_prev[0], lx + (X1 * h * 0.5 + dlx) + g_prev[1])       # :100, :286
    return lam_new, th0


def _analytic_nve_adjoint(func, t, ans, grad_output, flat_params):
    """OdeintAdjointMethod.backward for `verlet` over an integrator with rhs_vjp, without the 6-tuple solver algebra: per
    interval the counter-only call and two analytic force-vjp evaluations (`nve_adjoint_interval`); replayed from a HIP
    graph where that applies.  The parameter term of an interval is  theta_vjp(x_i; lam_v) * h  (sovlers.py:82, :101: the half
    step taken twice)."""
    with torch.no_grad():
        T = ans[0].shape[0]

        def eager():
            lam = tuple(g[-1].clone() for g in grad_output)
            gth = torch.zeros_like(flat_params)
            for i in range(T - 1, 0, -1):
                h = t[i] - t[i - 1]
                lam, th0 = nve_adjoint_interval(func, ans[0][i], ans[1][i], lam, h, (grad_output[0][i - 1], grad_output[1][i - 1]))
                if th0:
                    gth = gth + _flatten(th0) * 0.5 * h * 2        # :82, :101 (the half step taken twice)
            return list(lam), gth

        out = None
        if graphs.enabled(func) and T > 3:
            out = graphs.adjoint(func, t, ans, grad_output, flat_params.numel())
            if out is None:
                out = graphs.eager_static(func, eager, lambda o: ans[1][0])
        if out is None:
            out = eager()
        return (*out[0], None, None, out[1], None, None, None, None, None)


def _fused_spec_for(func, method):
    get = getattr(func, "fused_spec", None)
    if get is None:
        return None
    return get(method)


def odeint_adjoint(func, y0, t, rtol=1e-6, atol=1e-12, method=None, options=None):
    """sovlers.py:296-324.  Dispatches to the fused HIP trajectory when `func` allows it."""
    if not isinstance(func, nn.Module):
        raise ValueError('func is required to be an instance of nn.Module.')
    spec = _fused_spec_for(func, method) if (isinstance(y0, (tuple, list)) and not options) else None
    if spec is not None:
        flat_params = spec.flat_params()
        y0 = tuple(y0)
        pv0 = y0[2] if spec.ensemble == 0 else None
        R, N = getattr(spec, "n_rep", 1), spec.n_atoms
        if R > 1 and y0[0].dim() == 2:
            # replica-stacked system ([R*N, 3] states): one workgroup (or grid row) per replica
            spec._fuse_now = True
            try:
                outs = ops.FusedTrajFn.apply(y0[0].reshape(R, N, 3), y0[1].reshape(R, N, 3),
                                             pv0.reshape(R, -1) if pv0 is not None else None, t, flat_params, spec)
            finally:
                spec._fuse_now = False
            T_ = t.shape[0]
            res = [outs[0].transpose(0, 1).reshape(T_, R * N, 3), outs[1].transpose(0, 1).reshape(T_, R * N, 3)]
            if pv0 is not None:
                res.append(outs[2].transpose(0, 1))
            n = 3 if pv0 is not None else 2
            ops.tag_trajectory(res[1], spec, outs[n] if len(outs) > n else None, 0)
            return tuple(res)
        return ops.fused_traj(y0[0], y0[1], pv0, t, flat_params, spec)

    single = torch.is_tensor(y0)
    if single:
        func, y0 = _AsTuple(func), (y0,)                   # a bare tensor state travels as a 1-tuple
    flat_params = _flatten(func.parameters())
    ys = OdeintAdjointMethod.apply(*y0, func, t, flat_params, rtol, atol, method, options)
    return ys[0] if single else ys


class _AsTuple(nn.Module):
    """Adapter for a right-hand side written for a single tensor state."""

    def __init__(self, rhs):
        super().__init__()
        self.rhs = rhs

    def forward(self, t, y):
        return (self.rhs(t, y[0]),)
