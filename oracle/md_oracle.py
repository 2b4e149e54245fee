"""CPU restatement of the pair-potential MD hot path.  TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py).  All citations are relative to /root/reference.

Derivatives are written out analytically (phi', phi'', d phi'/d theta) instead
of autograd-of-autograd; the *algorithm* (call pattern, update order, the
reference's forward-sign midpoint in the adjoint, the topology counter) is the
reference's.  dtype follows the inputs (fp32 for parity, fp64 for calibration).
"""
import math

import numpy as np
import torch

__all__ = [
    "cell_matrix", "nbr_list", "compute_dis", "pair_phi", "PairTerm", "ModelOracle", "BondTerm", "AngleTerm",
    "get_offsets_oracle",
    "NHCOracle", "NVEOracle", "odeint_oracle", "adjoint_oracle", "rdf_oracle", "rdf_raw_oracle",
    "rdf_normalise_oracle", "vacf_oracle", "temperature_oracle",
    "vol_bins_oracle", "wrap_positions_oracle", "fcc_lattice", "diamond_lattice",
]


# --------------------------------------------------------------------------- topology
def cell_matrix(cell):
    """1-D cell -> diagonal matrix (torchmd/topology.py:55-56)."""
    cell = torch.as_tensor(cell)
    return torch.diag(cell) if cell.dim() == 1 else cell


def _pair_select_mask(N, index_tuple, ex_pairs):
    """Boolean [N,N] selection equivalent to the reference's multiplicative 0/1 masks
    (torchmd/topology.py:15-27, 37-53): masked pairs get D=0 and are dropped by d2 != 0."""
    keep = torch.ones(N, N, dtype=torch.bool)
    if index_tuple is not None:
        a = torch.as_tensor(list(index_tuple[0]), dtype=torch.long)
        b = torch.as_tensor(list(index_tuple[1]), dtype=torch.long)
        sel = torch.zeros(N, N, dtype=torch.bool)
        sel[a[:, None], b[None, :]] = True
        sel = sel | sel.t()
        keep &= sel
    if ex_pairs is not None:
        ex = torch.as_tensor(ex_pairs, dtype=torch.long).reshape(-1, 2)
        keep[ex[:, 0], ex[:, 1]] = False
        keep[ex[:, 1], ex[:, 0]] = False
    return keep


def nbr_list(xyz, cutoff, cell, index_tuple=None, ex_pairs=None, get_dis=False):
    """Minimum-image half neighbour list (torchmd/topology.py:30-73).

    Pairs are enumerated explicitly as i<j (row-major), which is the order
    `torch.nonzero(triu(mask))` yields; batched input [...,N,3] prepends the frame
    index like the reference.  Returns (nbr, offsets[, dis]); offsets in {-1,0,1}.
    """
    xyz = torch.as_tensor(xyz)
    cellm = cell_matrix(cell).to(xyz)
    N = xyz.shape[-2]
    iu = torch.triu_indices(N, N, offset=1)
    i, j = iu[0], iu[1]
    D = xyz[..., j, :] - xyz[..., i, :]                       # :35  x_j - x_i
    keep_sel = _pair_select_mask(N, index_tuple, ex_pairs)[i, j]
    D = D * keep_sel[..., None].to(D)                         # :40-53 masked -> 0
    s = D.matmul(cellm.inverse())                             # :59
    o = -(s > 0.5).to(D) + (s < -0.5).to(D)                   # :60-62 strict tests
    D = D + o.matmul(cellm)                                   # :64
    d2 = D.pow(2).sum(-1)                                     # :66
    mask = (d2 < cutoff ** 2) & (d2 != 0)                     # :67
    if xyz.dim() == 2:
        idx = torch.nonzero(mask, as_tuple=False)[:, 0]
        nbr = torch.stack([i[idx], j[idx]], dim=1)
        off = o[idx]
        dis = d2[idx].sqrt()
    else:
        lead = mask.shape[:-1]
        nz = torch.nonzero(mask.reshape(-1, mask.shape[-1]), as_tuple=False)
        fr, idx = nz[:, 0], nz[:, 1]
        nbr = torch.stack([fr, i[idx], j[idx]], dim=1)
        off = o.reshape(-1, o.shape[-2], 3)[fr, idx]
        dis = d2.reshape(-1, d2.shape[-1])[fr, idx].sqrt()
        del lead
    if get_dis:
        return nbr, dis, off
    return nbr, off


def compute_dis(xyz, nbr, offsets, cell):
    """torchmd/topology.py:5-12 -- returns (d[P,3], r[P])."""
    d = xyz[nbr[:, 0]] - xyz[nbr[:, 1]] - offsets.matmul(cell_matrix(cell).to(xyz))
    return d, d.pow(2).sum(1).sqrt()


# --------------------------------------------------------------------------- pair forms
def _ipow(x, n):
    return x ** int(n)


def pair_phi(kind, r, theta, consts):
    """phi(r) and derivatives.  Returns (u, du, d2u, du_dtheta[K], ddu_dtheta[K]) with
    du = dphi/dr, d2u = d2phi/dr2, ddu_dtheta = d(dphi/dr)/dtheta_k.

    kinds / reference forms:
      'lj'     4 eps [(s/r)^p - c (s/r)^q], theta=(sigma, eps); covers LennardJones (:317-327,
               p=12,q=6,c=1), LJFamily (:61-73), LennardJones69 (:329-339, p=9),
               ExcludedVolume (:341-352, c=0)                     torchmd/potentials.py
      'morse'  ModifiedMorse (:75-93): x=a(1-r^phi)/phi, u=(e^{2x}-2e^x-A)/(1+A); no params
      'buck'   Buck (:354-365): A e^{-B r} - C / r^6, theta=(A,B,C)
      'yukawa' eps e^{-kappa r}/r, theta=(eps,kappa)  -- NOT in the reference (parity unpinned)
    """
    if kind == "lj":
        p, q, c = consts["p"], consts["q"], consts["c"]
        sig, eps = theta[0], theta[1]
        s = sig / r
        sp = _ipow(s, p)
        sq = _ipow(s, q) * c if c != 0 else torch.zeros_like(r)
        u = 4 * eps * (sp - sq)
        du = 4 * eps * (-p * sp + q * sq) / r
        d2u = 4 * eps * (p * (p + 1) * sp - q * (q + 1) * sq) / (r * r)
        du_dth = [4 * eps * (p * sp - q * sq) / sig, 4 * (sp - sq)]
        ddu_dth = [4 * eps * (-p * p * sp + q * q * sq) / (r * sig), 4 * (-p * sp + q * sq) / r]
        return u, du, d2u, du_dth, ddu_dth
    if kind == "morse":
        a, ph = consts["a"], consts["phi"]
        A = 0.0 if ph >= 0 else math.exp(2 * a / ph) - 2 * math.exp(a / ph)
        rp = r ** ph
        x = a * (1 - rp) / ph
        e1, e2 = torch.exp(x), torch.exp(2 * x)
        u = (e2 - 2 * e1 - A) / (1 + A)
        ux = (2 * e2 - 2 * e1) / (1 + A)
        uxx = (4 * e2 - 2 * e1) / (1 + A)
        xr = -a * rp / r
        xrr = -a * (ph - 1) * rp / (r * r)
        return u, ux * xr, uxx * xr * xr + ux * xrr, [], []
    if kind == "buck":
        A, B, C = theta[0], theta[1], theta[2]
        e = torch.exp(-B * r)
        r6 = r ** 6
        u = A * e - C / r6
        du = -A * B * e + 6 * C / (r6 * r)
        d2u = A * B * B * e - 42 * C / (r6 * r * r)
        du_dth = [e, -A * r * e, -1 / r6]
        ddu_dth = [-B * e, -A * e + A * B * r * e, 6 / (r6 * r)]
        return u, du, d2u, du_dth, ddu_dth
    if kind == "yukawa":
        eps, kap = theta[0], theta[1]
        e = torch.exp(-kap * r)
        u = eps * e / r
        du = -eps * e * (kap * r + 1) / (r * r)
        d2u = eps * e * (kap * kap * r * r + 2 * kap * r + 2) / (r * r * r)
        du_dth = [e / r, -eps * e]
        ddu_dth = [-e * (kap * r + 1) / (r * r), eps * e * kap]
        return u, du, d2u, du_dth, ddu_dth
    raise ValueError(kind)


class PairTerm:
    """One PairPotentials term (torchmd/interface.py:217-300)."""

    def __init__(self, kind, theta, cutoff, cell, index_tuple=None, ex_pairs=None, **consts):
        self.kind, self.cutoff, self.consts = kind, float(cutoff), consts
        self.theta = torch.as_tensor(theta).reshape(-1)
        self.cell = cell_matrix(cell)
        self.index_tuple, self.ex_pairs = index_tuple, ex_pairs
        self.nbr = self.off = None

    @property
    def n_theta(self):
        return self.theta.numel()

    def reset(self, q):                                       # interface.py:263-282
        self.nbr, self.off = nbr_list(q.detach(), self.cutoff, self.cell.to(q), self.index_tuple,
                                      self.ex_pairs)

    def _geom(self, q):
        d, r = compute_dis(q, self.nbr, self.off.to(q), self.cell.to(q))
        return d, r, d / r[:, None]

    def energy(self, q):                                      # interface.py:298-300
        _, r, _ = self._geom(q)
        return pair_phi(self.kind, r, self.theta.to(q), self.consts)[0].sum()

    def force(self, q):
        d, r, rh = self._geom(q)
        du = pair_phi(self.kind, r, self.theta.to(q), self.consts)[1]
        t = du[:, None] * rh
        F = torch.zeros_like(q)
        F.index_add_(0, self.nbr[:, 0], -t)
        F.index_add_(0, self.nbr[:, 1], t)
        return F

    def force_vjp(self, q, w):
        """F, d(w.F)/dq (= -H w) and d(w.F)/dtheta  (what double-autograd yields at
        torchmd/sovlers.py:229-233)."""
        d, r, rh = self._geom(q)
        _, du, d2u, _, ddu_dth = pair_phi(self.kind, r, self.theta.to(q), self.consts)
        i, j = self.nbr[:, 0], self.nbr[:, 1]
        t = du[:, None] * rh
        F = torch.zeros_like(q)
        F.index_add_(0, i, -t)
        F.index_add_(0, j, t)
        wij = w[i] - w[j]
        a = (rh * wij).sum(1)
        hv = (d2u * a)[:, None] * rh + (du / r)[:, None] * (wij - a[:, None] * rh)
        dq = torch.zeros_like(q)
        dq.index_add_(0, i, -hv)
        dq.index_add_(0, j, hv)
        dth = (torch.stack([-(x * a).sum() for x in ddu_dth]) if ddu_dth
               else torch.zeros(0, dtype=q.dtype))
        return F, dq, dth


def get_offsets_oracle(vecs, cell_len):
    """topology.get_offsets (torchmd/topology.py:75-80): -[b >= L/2] + [b < -L/2] -- NON-strict on the upper side,
    unlike the neighbour list's test (:60-61)."""
    return -(vecs >= 0.5 * cell_len).to(vecs) + (vecs < -0.5 * cell_len).to(vecs)


class _BondedTerm:
    """A bonded term over a static topology table as an oracle term: no neighbour list, no parameters; force and the
    vjp d(w.F)/dq by autograd on the restated energy (double backward, like the reference at sovlers.py:229-233)."""

    n_theta = 0

    def __init__(self, top, k, x0, cell):
        self.top = torch.as_tensor(top, dtype=torch.long)
        self.k, self.x0 = float(k), float(x0)
        c = torch.as_tensor(cell, dtype=torch.float32)
        self.cell_len = torch.diag(c) if c.dim() == 2 else c        # interface.py:429-431: cell.diag()

    def reset(self, q):                                             # interface.py:439-440: pass
        pass

    def force(self, q):
        with torch.enable_grad():
            x = q.detach().requires_grad_(True)
            (g,) = torch.autograd.grad(self.energy(x), x)
        return -g

    def force_vjp(self, q, w):
        with torch.enable_grad():
            x = q.detach().requires_grad_(True)
            (g,) = torch.autograd.grad(self.energy(x), x, create_graph=True)
            (dq,) = torch.autograd.grad((w.detach() * (-g)).sum(), x)
        return (-g).detach(), dq.detach(), torch.zeros(0, dtype=q.dtype)


class BondTerm(_BondedTerm):
    """BondPotentials (torchmd/interface.py:406-456): 1/2 k (|b|^2 - ro)^2 -- harmonic in the SQUARED length."""

    def energy(self, q):
        L = self.cell_len.to(q)
        b = q[self.top[:, 0]] - q[self.top[:, 1]]                   # :447
        b = b + get_offsets_oracle(b, L) * L                        # :448-449
        return 0.5 * self.k * (b.pow(2).sum(-1) - self.x0).pow(2).sum(-1)   # :450-453


class AngleTerm(_BondedTerm):
    """AnglePotentials (torchmd/interface.py:457-510): 1/2 k (theta - theta0)^2, triples (i, j, k) centred on j."""

    def energy(self, q):
        L = self.cell_len.to(q)
        b1 = q[self.top[:, 0]] - q[self.top[:, 1]]                  # :496
        b2 = q[self.top[:, 2]] - q[self.top[:, 1]]                  # :497
        b1 = b1 + get_offsets_oracle(b1, L) * L
        b2 = b2 + get_offsets_oracle(b2, L) * L
        cos = (b1 * b2).sum(-1) / (b1.pow(2).sum(-1) * b2.pow(2).sum(-1)).sqrt()   # :500-503
        return 0.5 * self.k * (torch.acos(cos) - self.x0).pow(2).sum(-1)           # :505-507


class ModelOracle:
    """Stack of terms (torchmd/interface.py:364-403); theta order = term order."""

    def __init__(self, terms):
        self.terms = list(terms)

    @property
    def n_theta(self):
        return sum(t.n_theta for t in self.terms)

    def reset(self, q):
        for t in self.terms:
            t.reset(q)

    def energy(self, q):
        return sum(t.energy(q) for t in self.terms)

    def force(self, q):
        return sum(t.force(q) for t in self.terms)

    def force_vjp(self, q, w):
        F = torch.zeros_like(q)
        dq = torch.zeros_like(q)
        dth = []
        for t in self.terms:
            f, g, h = t.force_vjp(q, w)
            F, dq = F + f, dq + g
            dth.append(h.reshape(-1))
        return F, dq, (torch.cat(dth) if dth else torch.zeros(0, dtype=q.dtype))


# --------------------------------------------------------------------------- equations of motion
class _EOMBase:
    def __init__(self, model, freq):
        self.model, self.freq, self.update_count = model, int(freq), 0

    def update_topology(self, q):                             # md.py:200-204 / :127-131
        if self.update_count % self.freq == 0:
            self.model.reset(q)
        self.update_count += 1


class NHCOracle(_EOMBase):
    """NoseHooverChain right-hand side and its vjp (torchmd/md.py:159-249)."""
    n_state = 3

    def __init__(self, model, mass, T, Q, num_chains, dim=3, freq=1):
        super().__init__(model, freq)
        self.mass = torch.as_tensor(mass)
        N = self.mass.shape[0]
        self.T = float(T)
        self.N_dof = N * dim                                  # md.py:187
        Qv = np.array([Q] + [Q / N] * (num_chains - 1))       # md.py:191-193
        self.Q = torch.as_tensor(Qv).to(self.mass)
        self.C = num_chains

    def _bath(self, pv, ke):                                  # md.py:234-236
        Q, T = self.Q.to(pv), self.T
        d0 = 2 * (ke - T * self.N_dof * 0.5) - pv[0] * pv[1] / Q[1]
        mid = (pv[:-2].pow(2) / Q[:-2] - T) - pv[2:] * pv[1:-1] / Q[2:]
        last = pv[-2].pow(2) / Q[-2] - T
        return torch.cat((d0[None], mid, last[None]))

    def rhs(self, y, compute=True):
        v, q, pv = y
        self.update_topology(q)
        if not compute:
            return None
        m = self.mass.to(v)[:, None]
        p = v * m
        ke = 0.5 * (p.pow(2) / m).sum()                       # md.py:223
        F = self.model.force(q)
        dv = (F - pv[0] * p / self.Q.to(v)[0]) / m            # md.py:230-238
        return dv, v, self._bath(pv, ke)

    def rhs_vjp(self, y, lam):
        """f(y) and lam^T df/d(y, theta) -- SURVEY A.6c."""
        v, q, pv = y
        lv, lq, lp = lam
        self.update_topology(q)
        m = self.mass.to(v)[:, None]
        Q = self.Q.to(v)
        p = v * m
        ke = 0.5 * (p.pow(2) / m).sum()
        F, dwF_dq, dth = self.model.force_vjp(q, lv / m)
        dv = (F - pv[0] * p / Q[0]) / m
        dpv = self._bath(pv, ke)
        C = self.C
        Gv = -(pv[0] / Q[0]) * lv + lq + 2 * m * v * lp[0]
        Gq = dwF_dq
        Gp = torch.zeros_like(pv)
        Gp[0] = -(lv * v).sum() / Q[0] - lp[0] * pv[1] / Q[1] + 2 * pv[0] * lp[1] / Q[0]
        for k in range(1, C - 1):
            Gp[k] = (-lp[k - 1] * pv[k - 1] / Q[k] - lp[k] * pv[k + 1] / Q[k + 1]
                     + 2 * pv[k] * lp[k + 1] / Q[k])
        Gp[C - 1] = -lp[C - 2] * pv[C - 2] / Q[C - 1]
        return (dv, v, dpv), (Gv, Gq, Gp), dth


class NVEOracle(_EOMBase):
    """NVE right-hand side (torchmd/md.py:98-157); note dv/dt = F with NO 1/m (:145-148)."""
    n_state = 2

    def __init__(self, model, mass=None, freq=1):
        super().__init__(model, freq)

    def rhs(self, y, compute=True):
        v, q = y
        self.update_topology(q)
        if not compute:
            return None
        return self.model.force(q), v

    def rhs_vjp(self, y, lam):
        v, q = y
        lv, lq = lam
        self.update_topology(q)
        F, dwF_dq, dth = self.model.force_vjp(q, lv)
        return (F, v), (lq, dwF_dq), dth


# --------------------------------------------------------------------------- solvers
def _step_forward(eom, y, dt):
    """NHverlet_update / verlet_update forward branches (torchmd/sovlers.py:110-127, 25-40):
    two func calls per step, nothing cached."""
    if eom.n_state == 3:
        v, q, pv = y
        a0, _, b0 = eom.rhs(y)
        vh = 0.5 * a0 * dt
        ph = 0.5 * b0 * dt
        qs = (v + vh) * dt
        a1, _, b1 = eom.rhs((v + vh, q + qs, pv + ph))
        return (v + (vh + 0.5 * a1 * dt), q + qs, pv + (ph + 0.5 * b1 * dt))
    v, q = y
    a0, _ = eom.rhs(y)
    vh = 0.5 * a0 * dt
    qs = (v + vh) * dt
    a1, _ = eom.rhs((v + vh, q + qs))
    return (v + (vh + 0.5 * a1 * dt), q + qs)


def odeint_oracle(eom, y0, t):
    """FixedGridODESolver.integrate on the grid t (torchmd/tinydiffeq.py:56-76)."""
    t = torch.as_tensor(t).to(y0[0])
    sol = [tuple(y0)]
    y = tuple(y0)
    for k in range(len(t) - 1):
        y = _step_forward(eom, y, t[k + 1] - t[k])
        sol.append(y)
    return tuple(torch.stack([s[i] for s in sol]) for i in range(len(y0)))


def adjoint_oracle(eom, traj, grad_out, t):
    """OdeintAdjointMethod.backward (torchmd/sovlers.py:211-293) with the backward branches
    of NHverlet_update (:129-164) / verlet_update (:42-101).  Returns (adj_y0 tuple, adj_theta).
    grad_out[i] may be None (treated as zeros)."""
    t = torch.as_tensor(t).to(traj[0])
    n = len(traj)
    T = traj[0].shape[0]
    go = [g if g is not None else torch.zeros_like(x) for g, x in zip(grad_out, traj)]
    lam = tuple(g[-1].clone() for g in go)
    gth = torch.zeros(eom.model.n_theta, dtype=traj[0].dtype)
    for i in range(T - 1, 0, -1):
        y = tuple(x[i] for x in traj)
        h = t[i] - t[i - 1]
        eom.rhs(y, compute=False)                             # :258 dL/dt call (counter only)
        if n == 3:
            (a, _, b), G0, th0 = eom.rhs_vjp(y, lam)
            v, q, pv = y
            vh = 0.5 * (-a) * h                               # :132
            ph = 0.5 * (-b) * h                               # :135
            qs = (v + vh) * h                                 # :138  (forward-time sign)
            lam_h = tuple(l + g * 0.5 * h for l, g in zip(lam, G0))
            _, G1, th1 = eom.rhs_vjp((v + vh, q + qs, pv + ph), lam_h)
            lam = tuple(l + g * h for l, g in zip(lam, G1))   # :156-160
            gth = gth + th1 * h
            # (the reference evaluates theta-adjoint at gth + th0*h/2 but only adds th1*h)
        else:
            v, x = y
            lv, lx = lam
            (F, _), (_, X0), th0 = eom.rhs_vjp(y, lam)
            dv = -F
            vhalf = v - 0.5 * dv * h                          # :49-50
            x0 = x - vhalf * h                                # :51-52
            dx = X0 * h * 0.5                                 # :71
            dvad = (lx + dx) * h                              # :72
            _, (_, X1), _ = eom.rhs_vjp((vhalf, x0), (lv + dvad, lx + dx))
            lam = (lv + dvad, lx + (X1 * h * 0.5 + dx))       # :100
            gth = gth + (th0 * 0.5 * h) * 2                   # :82,101
        lam = tuple(l + g[i - 1] for l, g in zip(lam, go))    # :286
    return lam, gth


# --------------------------------------------------------------------------- observables
def vol_bins_oracle(start, end, nbins, dim=3):
    """generate_vol_bins (torchmd/observable.py:10-21)."""
    bins = torch.linspace(start, end, nbins + 1)
    if dim == 3:
        vb = 4 * np.pi / 3 * (bins[1:] ** 3 - bins[:-1] ** 3)
        V = (4 / 3) * np.pi * end ** 3
    else:
        vb = np.pi * (bins[1:] ** 2 - bins[:-1] ** 2)
        V = np.pi * end ** 2
    return V, vb, bins


def rdf_raw_oracle(xyz, cell, nbins, r_range, index_tuple=None, width=None, dim=3):
    """The un-normalised soft histogram of rdf.forward: GaussianSmearing(pair distances).sum(0)
    (torchmd/observable.py:64-70, nff/nn/layers.py:14-31).  Additive over frames, so callers with many
    frames can accumulate it chunk by chunk."""
    start, end = r_range
    V, vb, bins = vol_bins_oracle(start, end, nbins, dim)
    mu = torch.linspace(start, float(bins[-1]), nbins).to(xyz)
    wd = (mu[1] - mu[0]) if width is None else torch.as_tensor(width).to(xyz)
    cellm = cell_matrix(cell).to(xyz)
    nbr, _, off = nbr_list(xyz.detach(), end + 0.5, cellm, index_tuple, get_dis=True)
    if xyz.dim() == 2:
        d = xyz[nbr[:, 0]] - xyz[nbr[:, 1]] - off.matmul(cellm)
    else:
        flat = xyz.reshape(-1, xyz.shape[-2], 3)
        d = flat[nbr[:, 0], nbr[:, 1]] - flat[nbr[:, 0], nbr[:, 2]] - off.matmul(cellm)
    r = d.pow(2).sum(-1).sqrt()
    coeff = -0.5 / wd ** 2
    return torch.exp(coeff * (r[:, None] - mu[None, :]) ** 2).sum(0)


def rdf_normalise_oracle(raw, nbins, r_range, dim=3):
    """count / g(r) from the raw histogram (torchmd/observable.py:71-76)."""
    V, vb, bins = vol_bins_oracle(r_range[0], r_range[1], nbins, dim)
    count = raw / raw.sum()
    return count, bins, count / (vb.to(raw) / V)


def rdf_oracle(xyz, cell, nbins, r_range, index_tuple=None, width=None, dim=3):
    """rdf.forward (torchmd/observable.py:33-76) with GaussianSmearing
    (nff/nn/layers.py:14-31,34-83).  Differentiable w.r.t. xyz through torch autograd."""
    raw = rdf_raw_oracle(xyz, cell, nbins, r_range, index_tuple, width, dim)
    return rdf_normalise_oracle(raw, nbins, r_range, dim)


def vacf_oracle(vel, t_range):
    """vacf.forward (torchmd/observable.py:153-163): <v(s+t) . v(s)> as the MEAN over frames, atoms and
    components for lags t = 0 .. t_range-1; vel [T, N, 3]."""
    out = [(vel * vel).mean()[None]]
    out += [(vel[t:] * vel[:-t]).mean()[None] for t in range(1, t_range)]
    return torch.stack(out).reshape(-1)


def temperature_oracle(vel, mass, dim=3):
    """Temperature.forward (torchmd/thermo.py:57-66) of one frame vel [N, 3]: 2 KE / N_dof, N_dof = N dim,
    in energy units."""
    p = vel * mass[:, None]
    ke = 0.5 * (p.pow(2) / mass[:, None]).sum()
    return ke / (mass.shape[0] * dim * 0.5)


# --------------------------------------------------------------------------- host utilities
def wrap_positions_oracle(pos, cell, eps=1e-7):
    """ase.geometry.wrap_positions as used at torchmd/md.py:66 (ase 3.20.1 is a third-party
    dependency absent from the tree: restated from its documented behaviour; parity unpinned)."""
    cell = np.asarray(cell_matrix(torch.as_tensor(np.asarray(cell, dtype=np.float64))))
    shift = -eps
    frac = np.linalg.solve(cell.T, np.asarray(pos, dtype=np.float64).T).T - shift
    frac %= 1.0
    frac += shift
    return frac @ cell


def _lattice(size, a, basis):
    pts = [(np.array([i, j, k]) + b) * a
           for i in range(size) for j in range(size) for k in range(size) for b in basis]
    return np.array(pts), np.array([a * size] * 3)


def fcc_lattice(size, a):
    return _lattice(size, a, np.array([[0, 0, 0], [.5, .5, 0], [.5, 0, .5], [0, .5, .5]]))


def diamond_lattice(size, a):
    return _lattice(size, a, np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0],
                                       [.25, .25, .25], [.25, .75, .75], [.75, .25, .75],
                                       [.75, .75, .25]]))
