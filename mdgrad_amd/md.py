"""Simulation driver and equations of motion with the reference's interface
(torchmd/md.py): Simulations :14-96, NVE :98-157, NoseHooverChain :159-249.

`forward(t, state)` is the generic right-hand side (torch ops around the HIP energy ops,
differentiable twice) that the reference's Python solvers call; `fused_spec(method)` describes
the same dynamics to the fused HIP trajectory kernels when that is possible.
"""
import numpy as np
import weakref

import torch

from . import units, ops
from .interface import PairPotentials, Stack
from .sovlers import odeint_adjoint, odeint
from .system import wrap_positions
from .tinydiffeq import _flatten

FUSED_MAX_ATOMS = 1024          # one workgroup per replica (csrc/traj_small.hip)
FUSED_MAX_ATOMS_LARGE = 32768   # multi-launch kernels (csrc/traj_large.hip), NoseHooverChain and NVE


def compute_grad(inputs, output, create_graph=True, retain_graph=True):
    """nff/utils/scatter.py:5-21."""
    assert inputs.requires_grad
    (g,) = torch.autograd.grad(output, inputs, grad_outputs=output.data.new(output.shape).fill_(1),
                               create_graph=create_graph, retain_graph=retain_graph)
    return g



def _to_device_f32(var, device):
    """torch.Tensor(var).to(device) -- the same float32 rounding -- through numpy: torch.Tensor(float64 ndarray) converts
    element by element (20 ms for the 32 768 x 3 positions of eight stacked 4 096-bead systems; this is 0.03 ms), and the call
    sits inside every pass of a training loop (demo/fit_rdf_gnn.py:405-408)."""
    return torch.from_numpy(np.ascontiguousarray(np.asarray(var, dtype=np.float64).astype(np.float32))).to(device)


class Simulations():
    """Epoch driver with the reference's interface and bookkeeping (torchmd/md.py:14-96).  One call of `simulate` runs
    steps // frequency epochs; an epoch integrates `frequency` time points (so frequency - 1 steps), appends
    its LAST frame of every state variable to `log` (host numpy, keyed by the integrator's `state_keys`), pushes positions
    and velocities back into the System, and the next epoch starts from that frame -- positions wrapped into the cell when
    `wrap` is set.  Only the final epoch's trajectories are returned, so only they carry gradient."""

    def __init__(self, system, integrator, wrap=True, method="NH_verlet"):
        self.system, self.integrator = system, integrator
        self.device = system.device
        self.solvemethod, self.wrap = method, wrap
        self.keys = integrator.state_keys
        self.initialize_log()

    # ---- the log: one list of last frames per state variable
    def initialize_log(self):
        self.log = dict((name, []) for name in self.keys)

    def update_log(self, trajs):
        for name, traj in zip(self.keys, trajs):
            self.log[name].append(traj[-1].detach().cpu().numpy())

    def update_states(self):
        for name, setter in (("positions", self.system.set_positions), ("velocities", self.system.set_velocities)):
            if name in self.log:
                setter(self.log[name][-1])

    def get_check_point(self):
        """Device tensors of the newest logged frame, in `keys` order; index 1 (the positions) wrapped when `wrap`."""
        log = getattr(self, "log", None)
        if log is None:
            raise ValueError("No log available")
        newest = [torch.Tensor(frames[-1]).to(self.device) for frames in log.values()]
        if self.wrap:
            newest[1] = torch.Tensor(wrap_positions(log["positions"][-1], self.system.get_cell())).to(self.device)
        return newest

    def _integrate(self, states, t):
        if self.integrator.adjoint:
            return odeint_adjoint(self.integrator, states, t, method=self.solvemethod)
        for x in states:                                    # (the reference's plain-autograd branch, md.py:88-91)
            x.requires_grad = True
        return odeint(self.integrator, tuple(states), t, method=self.solvemethod)

    def simulate(self, steps=1, dt=1.0 * units.fs, frequency=1):
        fresh = len(self.log["positions"]) == 0
        states = self.integrator.get_inital_states(self.wrap) if fresh else self.get_check_point()
        t = torch.Tensor([dt * k for k in range(frequency)]).to(self.device)
        trajs = None
        for _ in range(int(steps // frequency)):
            trajs = self._integrate(states, t)
            self.update_log(trajs)
            self.update_states()
            states = self.get_check_point()
        return trajs


def _pair_terms_of(model):
    """PairPotentials terms of `model` in parameter order, or None when something else
    (a GNN, a user module, a non-built-in pair form) takes part."""
    if isinstance(model, PairPotentials):
        mods = [model]
    elif isinstance(model, Stack):
        mods = list(model.models.values())
    else:
        return None
    for m in mods:
        if not isinstance(m, PairPotentials) or not m.builtin():
            return None
    return mods if 1 <= len(mods) <= 4 else None


class _FusedSpec(ops.FusedSpec):
    def __init__(self, integrator, *a, **k):
        super().__init__(*a, **k)
        self._integrator = integrator

    def flat_params(self):
        return _flatten(self._integrator.parameters())          # sovlers.py:319


def _table_members(model):
    """Members of a (Stack of) PairPotentials of which at least one is a user module (pairMLP ...), all
    unmasked with one cutoff: the whole pair energy is then tabulated for the fused kernels."""
    mods = [model] if isinstance(model, PairPotentials) else (list(model.models.values()) if isinstance(model, Stack) else None)
    if not mods or not all(isinstance(m, PairPotentials) for m in mods):
        return None
    if all(m.builtin() for m in mods) or any(m._mask is not None for m in mods):
        return None
    if len({float(m.cutoff) for m in mods}) != 1 or not mods[0]._cell_struct.diag:
        return None
    return mods


class _TableSpec(ops.FusedSpec):
    """FusedSpec whose single term is MDG_PAIR_TABLE: `flat_params()` evaluates c1(u) = phi'(r)/r = 2 dphi/du
    and its slope on the uniform u = r^2 grid with autograd (differentiable w.r.t. every module parameter),
    so the table gradient the adjoint kernel returns flows on into the modules."""

    def __init__(self, integrator, members, nodes, r_min, *a, **k):
        super().__init__(*a, **k)
        self._integrator, self._members = integrator, members
        rc = float(members[0].cutoff)
        self.u0, self.nodes = float(r_min) ** 2, int(nodes)
        self.du = (rc * rc - self.u0) / (self.nodes - 1)
        self.table = True

    def flat_params(self):
        dev = self._integrator.mass.device
        with torch.enable_grad():
            u = (self.u0 + self.du * torch.arange(self.nodes, device=dev, dtype=torch.float32)).requires_grad_(True)
            r = u.sqrt()[:, None]
            phi = sum(m._phi(r).reshape(-1) for m in self._members)
            (dphi,) = torch.autograd.grad(phi.sum(), u, create_graph=True)
            c1 = 2.0 * dphi                                          # phi'(r) / r
            (dc1,) = torch.autograd.grad(c1.sum(), u, create_graph=True)
            return torch.stack((c1, self.du * dc1), 1).reshape(-1)


class _EOM(torch.nn.Module):
    _ensemble = None
    _method = None
    fused_large = None      # None = by size; True/False forces the multi-launch / one-workgroup kernels
    fused_table = True      # tabulate user pair modules for the fused kernels (N <= 1024); False: generic path
    fuse_observables = False   # True (or attach_observable): an rdf called on a fused trajectory is evaluated INSIDE the next
    #                            trajectory launches (its histogram is then an output of the launch, not a function of q_t in
    #                            the autograd graph: autograd.grad(loss, q_t) / hooks on q_t do not see the RDF term).  Off by
    #                            default: the reference's callers may rely on q_t carrying that term.
    table_nodes = 2048      # nodes of the u = r^2 grid on [ (table_rmin * cutoff)^2 , cutoff^2 ] (round 6: 1 024 -> 2 048, the
    #                         most the wave-per-replica kernels hold in LDS: ELU-module gradients to 6e-4 of the largest entry
    #                         instead of 1.1e-3, 1 % of the tabulated ring rate)
    table_rmin = 0.2

    def update_topology(self, q):                               # md.py:200-204
        freq = self.topology_update_freq
        if freq != 1:
            #  the fused stale-list kernels keep their lists in _stale_code, this path in the model's own
            # nbr_list: a pass of one kind between two rebuilds must not leave the other evaluating with older lists.  A
            # generic call drops the fused lists (the next fused pass waits for a rebuild count, fused_spec); the first
            # generic call after fused passes rebuilds here at once -- the reference's lists of that moment were built
            # inside the fused launch and exist only in its encoding.
            self._stale_code = None
            if getattr(self, "_stale_fused_dirty", False):
                self._stale_fused_dirty = False
                self.model._reset_topology(q)
                self._topo_ref = None
        if self.update_count % freq == 0:
            # The reference rebuilds here unconditionally; its adjoint asks twice in a row for the list at the
            # very same saved frame (the dL/dt call of sovlers.py:258, then the first augmented evaluation).  A
            # rebuild at the same tensor object (unchanged version, nobody else touched the model's topology in
            # between) returns the same list, so it is skipped; the counter still advances.
            # (held through a weak reference: `q` is usually a view of the saved trajectory, which must not be kept
            # alive past backward; inference-mode tensors have no version counter and always rebuild)
            m = self.model
            try:
                ver = q._version
            except RuntimeError:
                ver = None
            ref = getattr(self, "_topo_ref", None)
            same = (ver is not None and ref is not None and ref() is q and self._topo_ver == ver
                    and getattr(m, "_topo_stamp", None) is self._topo_stamp)
            if not same:
                m._reset_topology(q)
                self._topo_ref, self._topo_ver, self._topo_stamp = weakref.ref(q), ver, getattr(m, "_topo_stamp", None)
        self.update_count += 1

    def force(self, q):
        """F(q) = -dU/dq with the topology update of md.py:225-228 (used by the generic solver to
        reuse the force between the second evaluation of step k and the first of step k+1 -- same q,
        bit-identical result; only when topology_update_freq == 1)."""
        if getattr(self.model, "supports_force_vjp", lambda: False)():
            self.update_topology(q)
            return self.model.force(q)
        with torch.set_grad_enabled(True):
            q = q.detach().requires_grad_(True)
            self.update_topology(q)
            u = self.model(q)
            (g,) = torch.autograd.grad(u.sum(), q)
        return -g

    def attach_observable(self, obs, start=0, stride=1):
        """Ask the fused trajectory launches of this integrator to evaluate `obs` (an `observable.rdf`) on the frames
        start, start + stride, ... of every trajectory from now on -- what the observable otherwise arranges itself
        the first time it is called on (a time slice of) a fused trajectory.  `attach_observable(None)` detaches;
        `fuse_observables = False` switches the mechanism off.  Attaching IS the opt-in: by default
        (`fuse_observables = False`) an rdf stays a function of q_t in the autograd graph, exactly as in the reference."""
        self._rdf_hint = None if obs is None else ops.RdfFuse(obs, start, stride)
        if obs is not None:
            self.fuse_observables = True

    def fused_spec(self, method):
        """FusedSpec when the whole trajectory can run in the fused HIP kernels, else None."""
        if method != self._method or self.dim != 3:
            return None
        freq = int(self.topology_update_freq)
        mods = _pair_terms_of(self.model)
        N = getattr(self.system, "group_size", self.mass.shape[0])      # atoms per replica
        if freq != 1:
            # stale neighbour lists (md.py:200-204): the one-workgroup-per-replica kernels keep the lists of the last rebuild
            # and follow the reference's call counter (mdg_traj_*_small_stale); built-in pair forms, N <= FUSED_MAX_ATOMS, and
            # only from a call count at which the reference rebuilds too or with the lists of an earlier fused pass at hand
            # (round 6: beyond FUSED_MAX_ATOMS, or with fused_large = True, the launch-per-evaluation kernels with stale rows,
            #  mdg_traj_*_large_stale)
            stale_large = N > FUSED_MAX_ATOMS if self.fused_large is None else bool(self.fused_large)
            if (freq < 1 or mods is None or not self.adjoint or getattr(self, "fused_stale", True) is False
                    or N > (min(FUSED_MAX_ATOMS_LARGE, 32768) if stale_large else FUSED_MAX_ATOMS)):
                return None
            code = getattr(self, "_stale_code", None)
            shape = ((getattr(self.system, "n_replicas", 1) * N * 257,) if stale_large
                     else (getattr(self.system, "n_replicas", 1), N, N))
            if self.update_count % freq != 0 and (code is None or tuple(code.shape) != shape):
                return None                 # (between two rebuilds without the lists of this geometry: the generic path)
        table_large = N > FUSED_MAX_ATOMS if self.fused_large is None else bool(self.fused_large)
        if (mods is None and self.adjoint and self.fused_table
                and (N <= FUSED_MAX_ATOMS_LARGE if table_large else N <= FUSED_MAX_ATOMS)):
            members = _table_members(self.model)
            if members is not None and (self._ensemble == 1 or 2 <= self.num_chains <= 16):
                kw = {} if self._ensemble != 0 else dict(T=self.T, n_dof=self.N_dof, Q=[float(x) for x in self.Q.tolist()])
                nodes = int(self.table_nodes)
                spec = _TableSpec(self, members, nodes, self.table_rmin * float(members[0].cutoff), self._ensemble, N,
                                  self.mass[:N].contiguous(), members[0]._cell_struct, None, 2 * nodes, [None],
                                  large=table_large, **kw)
                desc = dict(kind=ops.MDG_PAIR_TABLE, p=nodes, a=spec.u0, phi=spec.du, c=1.0)
                spec.terms = ops.make_terms([ops.make_term(desc, members[0].cutoff, 0, 2 * nodes, None)], 2 * nodes)
                spec.n_rep = getattr(self.system, "n_replicas", 1)
                return spec
        if mods is None or not self.adjoint:
            return None
        large = N > FUSED_MAX_ATOMS if self.fused_large is None else bool(self.fused_large)
        if large and N > FUSED_MAX_ATOMS_LARGE:
            return None
        plist = list(self.parameters())
        offs, pos = {}, 0
        for p in plist:
            offs[id(p)] = pos
            pos += p.numel()
        terms, masks = [], []
        for m in mods:
            mp = m.model.mdg_params()
            off = offs[id(mp[0])] if mp else 0
            for a, b in zip(mp[:-1], mp[1:]):
                if offs[id(b)] != offs[id(a)] + a.numel():
                    return None                                   # parameters not contiguous
            terms.append(ops.make_term(m.model.mdg_term(), m.cutoff, off, sum(p.numel() for p in mp), m._mask))
            masks.append(m._mask)
        cs = mods[0]._cell_struct
        kw = {}
        if self._ensemble == 0:
            if not 2 <= self.num_chains <= 16:
                return None
            kw = dict(T=self.T, n_dof=self.N_dof, Q=[float(x) for x in self.Q.tolist()])
        spec = _FusedSpec(self, self._ensemble, N, self.mass[:N].contiguous(), cs, ops.make_terms(terms, pos), pos,
                          masks, large=large, **kw)
        spec.n_rep = getattr(self.system, "n_replicas", 1)
        spec.stale_freq = freq if freq != 1 else 0
        # an rdf observable that was evaluated on an earlier trajectory of this integrator (observable.py) is
        # computed inside the next fused launch when the caller opted in (`fuse_observables = True` / attach_observable)
        spec.rdf_hint = (getattr(self, "_rdf_hint", None) if (getattr(self, "fuse_observables", False) and freq == 1)
                         else None)
        return spec

    def stale_lists(self, n_rep, n_atoms, device, large=False):
        """Persistent neighbour-list buffer of the stale-list kernels ([R][N][N] uint16 words: pair set + image flags of
        every term as of the last rebuild; `large`: the rows of mdg_traj_*_large_stale, [R][N][256] uint32 entries + counts) --
        the fused counterpart of the reference's nbr_list / offsets attributes."""
        code = getattr(self, "_stale_code", None)
        shape = (n_rep * n_atoms * 257,) if large else (n_rep, n_atoms, n_atoms)
        if code is None or tuple(code.shape) != shape or code.device != device:
            if code is not None or self.update_count % int(self.topology_update_freq) != 0:
                raise RuntimeError("mdgrad_amd: the fused stale-list kernels have no lists for this launch (%s) while the call "
                                   "counter (%d, topology_update_freq %d) is between two rebuilds; set integrator.fused_stale = "
                                   "False" % ("another launch geometry holds them" if code is not None else
                                              "a generic force / odeint call on this integrator dropped them",
                                              self.update_count, int(self.topology_update_freq)))
            code = self._stale_code = torch.zeros(*shape, dtype=torch.int32 if large else torch.int16, device=device)
        return code


class NVE(_EOM):
    """torchmd/md.py:98-157 (dv/dt = F with no 1/m, :145-148)."""
    _ensemble, _method = 1, 'verlet'

    def __init__(self, potentials, system, adjoint=True, topology_update_freq=1):
        super().__init__()
        self.model = potentials
        self.system = system
        self.mass = torch.Tensor(system.get_masses()).to(self.system.device)
        self.N_dof = self.mass.shape[0] * system.dim
        self.dim = system.dim
        self.adjoint = adjoint
        self.state_keys = ['velocities', 'positions']
        self.topology_update_freq = topology_update_freq
        self.update_count = 0

    def forward(self, t, state):
        if not torch.is_grad_enabled() and self.supports_rhs_vjp():
            # graph-free callers (forward integration): the model's own force kernels, no autograd
            v, q = state[0], state[1]
            self.update_topology(q)
            return (self.model.force(q), v)
        with torch.set_grad_enabled(True):
            v, q = state[0], state[1]
            if self.adjoint:
                q.requires_grad = True
            self.update_topology(q)
            u = self.model(q)
            f = -compute_grad(inputs=q, output=u.sum(-1))
        return (f, v)

    # analytic-adjoint protocol (see NoseHooverChain.rhs_vjp): f = (F(q), v) -- no 1/m, md.py:145-148
    def supports_rhs_vjp(self):
        return getattr(self.model, "supports_force_vjp", lambda: False)()

    def rhs_vjp(self, state, adj, want_theta=True):
        v, q = state
        lv, lq = adj
        self.update_topology(q)
        F, dwF_dq, gth = self.model.force_vjp(q, lv, want_theta=want_theta)
        if not want_theta:
            return (F, v), (lq, dwF_dq), None
        by_id = {id(p): g for p, g in zip(self.model.parameters(), gth)}
        return (F, v), (lq, dwF_dq), [by_id[id(p)] if id(p) in by_id else torch.zeros_like(p)
                                      for p in self.parameters()]

    def get_inital_states(self, wrap=True):
        states = [self.system.get_velocities(), self.system.get_positions(wrap=wrap)]
        return [_to_device_f32(var, self.system.device) for var in states]


class NoseHooverChain(_EOM):
    """torchmd/md.py:159-249."""
    _ensemble, _method = 0, 'NH_verlet'

    def __init__(self, potentials, system, T, num_chains=2, Q=1.0, adjoint=True, topology_update_freq=1):
        super().__init__()
        self.model = potentials
        self.system = system
        self.device = system.device
        self.mass = torch.Tensor(system.get_masses()).to(self.device)
        self.T = T
        self.N_dof = self.mass.shape[0] * system.dim
        self.target_ke = (0.5 * self.N_dof * T)
        self.num_chains = num_chains
        # replica-stacked systems (System.replicate): one thermostat chain per replica of n_group atoms
        self.n_rep = getattr(system, "n_replicas", 1)
        self.n_group = getattr(system, "group_size", self.mass.shape[0])
        self.N_dof = self.n_group * system.dim
        self.target_ke = (0.5 * self.N_dof * T)
        self.Q = np.array([Q, *[Q / self.n_group] * (num_chains - 1)])
        self.Q = torch.Tensor(self.Q).to(self.device)
        self.dim = system.dim
        self.adjoint = adjoint
        self.state_keys = ['velocities', 'positions', 'baths']
        self.topology_update_freq = topology_update_freq
        self.update_count = 0

    @property
    def T(self):
        return self._T

    @T.setter
    def T(self, value):
        # the thermostat kernels (and the captured HIP graphs that replay them) read the temperature from a
        # device scalar: keep it in step with plain attribute assignment as well as update_T
        self._T = value
        if getattr(self, "_T_buf", None) is not None:
            self._T_device()

    def update_T(self, T):
        self.T = T
        self._T_device()

    def _T_device(self):
        """self.T mirrored in a device scalar (what the thermostat kernel reads)."""
        buf = getattr(self, "_T_buf", None)
        if buf is None or buf.device != self.mass.device:
            buf = self._T_buf = torch.empty(1, device=self.mass.device)
            self._T_val = None
        if self._T_val != float(self._T):
            self._T_val = float(self._T)
            buf.fill_(self._T_val)
        return buf

    def supports_rhs_vjp(self):
        return getattr(self.model, "supports_force_vjp", lambda: False)()

    def rhs_vjp(self, state, adj, want_theta=True):
        """f(y) and adj^T df/d(y, theta) written out analytically: what
        augmented_dynamics gets from autograd at torchmd/sovlers.py:229-233, with the model's
        force / Hessian-vector product / parameter vjp coming from its force_vjp (HIP kernels, no
        autograd graph).  Returns (f_eval tuple, vjp_y tuple, [vjp per parameter])."""
        v, q, p_v = state
        lv, lq, lp = adj
        self.update_topology(q)
        m = self.mass[:, None]
        F, dwF_dq, gth = self.model.force_vjp(q, lv / m, want_theta=want_theta)
        f_eval = self.rhs_from_force((v, q, p_v), F)
        Q = self.Q
        if self._hip_algebra(v, lv, lq, lp, p_v):
            Gv, Gp = ops.nhc_vjp(v, p_v, lv, lq, lp, self.mass, Q, self.n_rep, self.n_group)
        elif p_v.dim() == 1:
            pv0, lp0, slv = p_v[0], lp[0], (lv * v).sum()
            Gv = -(pv0 / Q[0]) * lv + lq + 2 * m * v * lp0
            Gp = torch.zeros_like(p_v)
            Gp[0] = -slv / Q[0] - lp[0] * p_v[1] / Q[1] + 2 * p_v[0] * lp[1] / Q[0]
            if p_v.shape[0] > 2:
                Gp[1:-1] = (-lp[:-2] * p_v[:-2] / Q[1:-1] - lp[1:-1] * p_v[2:] / Q[2:]
                            + 2 * p_v[1:-1] * lp[2:] / Q[1:-1])
            Gp[-1] = -lp[-2] * p_v[-2] / Q[-1]
        else:
            R, n = p_v.shape[0], self.n_group
            pv0 = p_v[:, 0].repeat_interleave(n)[:, None]
            lp0 = lp[:, 0].repeat_interleave(n)[:, None]
            slv = (lv * v).reshape(R, -1).sum(1)
            Gv = -(pv0 / Q[0]) * lv + lq + 2 * m * v * lp0
            Gp = torch.zeros_like(p_v)
            Gp[:, 0] = -slv / Q[0] - lp[:, 0] * p_v[:, 1] / Q[1] + 2 * p_v[:, 0] * lp[:, 1] / Q[0]
            if p_v.shape[1] > 2:
                Gp[:, 1:-1] = (-lp[:, :-2] * p_v[:, :-2] / Q[1:-1] - lp[:, 1:-1] * p_v[:, 2:] / Q[2:]
                               + 2 * p_v[:, 1:-1] * lp[:, 2:] / Q[1:-1])
            Gp[:, -1] = -lp[:, -2] * p_v[:, -2] / Q[-1]
        if not want_theta:
            return f_eval, (Gv, dwF_dq, Gp), None
        by_id = {id(p): g for p, g in zip(self.model.parameters(), gth)}
        return f_eval, (Gv, dwF_dq, Gp), [by_id[id(p)] if id(p) in by_id else torch.zeros_like(p)
                                         for p in self.parameters()]

    def fused_steps_ok(self, v, q, p_v):
        """The whole-half-step kernels (csrc/nhc.hip mdg_nhv_*) apply: analytic force / force-vjp from the model,
        contiguous fp32 device states, chain entries per replica."""
        return (getattr(self.model, "supports_force_vjp", lambda: False)() and self._hip_algebra(v, q, p_v)
                and v.is_contiguous() and q.is_contiguous() and p_v.is_contiguous()
                and 2 <= p_v.shape[-1] <= 16 and v.shape[0] == self.n_rep * self.n_group
                and getattr(self, "fused_steps", True))

    def nhv_work(self, v, p_v):
        w = getattr(self, "_nhv_work", None)
        if w is None or w.dv_h.shape != v.shape or w.dv_h.device != v.device or w.dp_h.shape != p_v.shape:
            w = self._nhv_work = ops.NhvWork(self, v, p_v)
        return w

    def theta_in_parameter_order(self, gth):
        """force_vjp's parameter gradients (model.parameters() order) as a list in self.parameters() order."""
        by_id = {id(p): g for p, g in zip(self.model.parameters(), gth)}
        return [by_id[id(p)] if id(p) in by_id else torch.zeros_like(p) for p in self.parameters()]

    def _hip_algebra(self, *tensors):
        """The single-launch thermostat kernels (csrc/nhc.hip) apply to fp32 device states outside
        autograd (the analytic-adjoint and graph-replay paths); autograd callers keep the torch ops."""
        if getattr(self, "hip_algebra", True) is False or self.dim != 3:
            return False
        if not all(t.is_cuda and t.dtype == torch.float32 for t in tensors):
            return False
        return not (torch.is_grad_enabled() and any(t.requires_grad for t in tensors))

    def rhs_from_force(self, state, f):
        """md.py:221-240 given F(q).  With R stacked replicas p_v is [R, C] and every replica has its
        own kinetic energy / friction (identical arithmetic per replica)."""
        v, q, p_v = state
        if self._hip_algebra(v, f, p_v):
            a, dpv = ops.nhc_rhs(v, f, p_v, self.mass, self.Q, self._T_device(), self.N_dof, self.n_rep, self.n_group)
            return (a, v, dpv)
        p = v * self.mass[:, None]
        if p_v.dim() == 1:
            sys_ke = 0.5 * (p.pow(2) / self.mass[:, None]).sum()
            coupled_forces = (p_v[0] * p.reshape(-1) / self.Q[0]).reshape(-1, 3)
            dpvdt_0 = 2 * (sys_ke - self.T * self.N_dof * 0.5) - p_v[0] * p_v[1] / self.Q[1]
            dpvdt_mid = (p_v[:-2].pow(2) / self.Q[:-2] - self.T) - p_v[2:] * p_v[1:-1] / self.Q[2:]
            dpvdt_last = p_v[-2].pow(2) / self.Q[-2] - self.T
            dpv = torch.cat((dpvdt_0[None], dpvdt_mid, dpvdt_last[None]))
        else:
            R = p_v.shape[0]
            sys_ke = 0.5 * (p.pow(2) / self.mass[:, None]).reshape(R, -1).sum(1)
            coupled_forces = (p_v[:, 0].repeat_interleave(self.n_group)[:, None] * p) / self.Q[0]
            dpvdt_0 = 2 * (sys_ke - self.T * self.N_dof * 0.5) - p_v[:, 0] * p_v[:, 1] / self.Q[1]
            dpvdt_mid = (p_v[:, :-2].pow(2) / self.Q[:-2] - self.T) - p_v[:, 2:] * p_v[:, 1:-1] / self.Q[2:]
            dpvdt_last = p_v[:, -2].pow(2) / self.Q[-2] - self.T
            dpv = torch.cat((dpvdt_0[:, None], dpvdt_mid, dpvdt_last[:, None]), 1)
        return ((f - coupled_forces) / self.mass[:, None], v, dpv)

    def forward(self, t, state):
        with torch.set_grad_enabled(True):
            v, q, p_v = state[0], state[1], state[2]
            if self.adjoint:
                q.requires_grad = True
            self.update_topology(q)
            u = self.model(q)
            f = -compute_grad(inputs=q, output=u.sum(-1))
            out = self.rhs_from_force((v, q, p_v), f)
        return out

    def get_inital_states(self, wrap=True):
        baths = [0.0] * self.num_chains if self.n_rep == 1 else np.zeros((self.n_rep, self.num_chains))
        states = [self.system.get_velocities(), self.system.get_positions(wrap=wrap), baths]
        return [_to_device_f32(var, self.system.device) for var in states]
