"""Pair functional forms with the reference's constructors and parameter names
(torchmd/potentials.py).  Each built-in form also describes itself to the HIP kernels through
`mdg_term()`; `forward(r)` evaluates phi(r) with torch ops for callers that use the module
directly (plots, tabulation) -- the MD hot path never calls it.
"""
import math

import torch
from torch import nn

from . import _lib


class _PairForm(nn.Module):
    """Base: a form the HIP kernels know.  mdg_term() -> dict(kind,p,q,c,a,phi), parameters in
    kernel order via mdg_params()."""

    def mdg_term(self):
        raise NotImplementedError

    def mdg_params(self):
        return []


class LJFamily(_PairForm):                                   # torchmd/potentials.py:61-73
    def __init__(self, sigma=1.0, epsilon=1.0, attr_pow=6, rep_pow=12):
        super().__init__()
        self.sigma = nn.Parameter(torch.Tensor([sigma]))
        self.epsilon = nn.Parameter(torch.Tensor([epsilon]))
        self.attr_pow = attr_pow
        self.rep_pow = rep_pow

    def LJ(self, r, sigma, epsilon):
        return 4 * epsilon * ((sigma / r) ** self.rep_pow - (sigma / r) ** self.attr_pow)

    def forward(self, x):
        return self.LJ(x, self.sigma, self.epsilon)

    def mdg_term(self):
        for v in (self.rep_pow, self.attr_pow):
            if int(v) != v or v < 0:
                raise ValueError("mdgrad_amd: LJ-family powers must be non-negative integers")
        return dict(kind=_lib.PAIR_LJ, p=int(self.rep_pow), q=int(self.attr_pow), c=1.0)

    def mdg_params(self):
        return [self.sigma, self.epsilon]


class LennardJones(LJFamily):                                # torchmd/potentials.py:317-327
    def __init__(self, sigma=1.0, epsilon=1.0):
        super().__init__(sigma, epsilon, attr_pow=6, rep_pow=12)


class LennardJones69(LJFamily):                              # torchmd/potentials.py:329-339
    def __init__(self, sigma=1.0, epsilon=1.0):
        super().__init__(sigma, epsilon, attr_pow=6, rep_pow=9)


class ExcludedVolume(_PairForm):                             # torchmd/potentials.py:341-352
    def __init__(self, sigma=1.0, epsilon=1.0, power=12):
        super().__init__()
        self.sigma = nn.Parameter(torch.Tensor([sigma]))
        self.epsilon = nn.Parameter(torch.Tensor([epsilon]))
        self.power = power

    def LJ(self, r, sigma, epsilon):
        return 4 * epsilon * ((sigma / r) ** self.power)

    def forward(self, x):
        return self.LJ(x, self.sigma, self.epsilon)

    def mdg_term(self):
        if int(self.power) != self.power or self.power < 1:
            raise ValueError("mdgrad_amd: ExcludedVolume power must be a positive integer")
        return dict(kind=_lib.PAIR_LJ, p=int(self.power), q=0, c=0.0)

    def mdg_params(self):
        return [self.sigma, self.epsilon]


class ModifiedMorse(_PairForm):                              # torchmd/potentials.py:75-93
    def __init__(self, a, phi):
        super().__init__()
        self.a = a
        self.phi = phi
        self.A = 0 if phi >= 0 else math.exp(2 * a / phi) - 2 * math.exp(a / phi)

    def forward(self, r):
        exponent = self.a * (1 - r ** self.phi) / self.phi
        return (torch.exp(2 * exponent) - 2 * torch.exp(exponent) - self.A) / (1 + self.A)

    def mdg_term(self):
        return dict(kind=_lib.PAIR_MORSE, a=float(self.a), phi=float(self.phi))


class Buck(_PairForm):                                       # torchmd/potentials.py:354-365
    def __init__(self, A=1.0, B=1.0, C=1.0):
        super().__init__()
        self.A = nn.Parameter(torch.Tensor([A]))
        self.B = nn.Parameter(torch.Tensor([B]))
        self.C = nn.Parameter(torch.Tensor([C]))

    def Buckingham(self, r, A, B, C):
        return A * torch.exp(-B * r) - C / r ** 6

    def forward(self, x):
        return self.Buckingham(x, self.A, self.B, self.C)

    def mdg_term(self):
        return dict(kind=_lib.PAIR_BUCK)

    def mdg_params(self):
        return [self.A, self.B, self.C]


class Yukawa(_PairForm):
    """u = epsilon * exp(-kappa r) / r.  Not in the reference (only Yukawa *data* exists,
    data/Yukawa_data); parity is unpinned and checked by finite differences."""

    def __init__(self, epsilon=1.0, kappa=1.0):
        super().__init__()
        self.epsilon = nn.Parameter(torch.Tensor([epsilon]))
        self.kappa = nn.Parameter(torch.Tensor([kappa]))

    def forward(self, x):
        return self.epsilon * torch.exp(-self.kappa * x) / x

    def mdg_term(self):
        return dict(kind=_lib.PAIR_YUKAWA)

    def mdg_params(self):
        return [self.epsilon, self.kappa]


def is_builtin_form(model):
    return isinstance(model, _PairForm)


# ----------------------------------------------------------------------------- per-pair neural potentials
# torchmd/potentials.py:13-22: the activation table of the LJ-fitting scripts (one shared module instance
# per name, as there)
nlr_dict = {name: getattr(nn, name)() for name in
            ("ReLU", "ELU", "Tanh", "LeakyReLU", "ReLU6", "SELU", "CELU", "Tanhshrink")}


class pairMLP(nn.Module):
    """phi(r) as an MLP over a trainable Gaussian expansion of the pair distance
    (torchmd/potentials.py:163-206; same layer order, so reference state_dicts load).  With `res`,
    width-preserving layers are residual.  Not a built-in kernel form: PairPotentials evaluates it per
    pair on the device from the HIP neighbour list (interface.PairPotentials module path)."""

    def __init__(self, n_gauss, r_start, r_end, n_layers, n_width, nonlinear, res=False):
        super().__init__()
        from .nn.layers import GaussianSmearing
        act = nlr_dict[nonlinear]
        self.smear = GaussianSmearing(start=r_start, stop=r_end, n_gaussians=n_gauss, trainable=True)
        widths = [n_gauss, n_gauss, n_width] + [n_width] * n_layers + [n_gauss]
        mods = []
        for a, b in zip(widths[:-1], widths[1:]):
            mods += [nn.Linear(a, b), act]
        mods.append(nn.Linear(n_gauss, 1))
        self.layers = nn.ModuleList(mods)
        self.res = res

    def forward(self, r):
        x = self.smear(r)
        for layer in self.layers:
            y = layer(x)
            x = x + y if (self.res and y.shape[-1] == x.shape[-1]) else y
        return x


class TpairMLP(nn.Module):
    """u(r, T) = energy(r) - T entropy(r), two pairMLPs (torchmd/potentials.py:208-217)."""

    def __init__(self, n_gauss, r_start, r_end, n_layers, n_width, nonlinear, res=False):
        super().__init__()
        self.energy = pairMLP(n_gauss, r_start, r_end, n_layers, n_width, nonlinear, res=res)
        self.entropy = pairMLP(n_gauss, r_start, r_end, n_layers, n_width, nonlinear, res=res)

    def forward(self, r, T):
        return self.energy(r) - T * self.entropy(r)
