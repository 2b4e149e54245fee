"""Observables with the reference's interface (torchmd/observable.py): generate_vol_bins
:10-21, Observable :24-31, rdf :33-76, vacf :153-163.  The pair search + Gaussian smearing +
histogram of rdf.forward is one HIP op (ops.RdfRawFn, csrc/rdf.hip); vacf is one fused reduction over the
velocity trajectory (ops.VacfFn, csrc/observe.hip)."""
import numpy as np
import torch

from . import _lib, ops
from .system import check_system


def generate_vol_bins(start, end, nbins, dim):
    bins = torch.linspace(start, end, nbins + 1)
    if dim == 3:
        Vbins = 4 * np.pi / 3 * (bins[1:] ** 3 - bins[:-1] ** 3)
        V = (4 / 3) * np.pi * (end) ** 3
    elif dim == 2:
        Vbins = np.pi * (bins[1:] ** 2 - bins[:-1] ** 2)
        V = np.pi * (end) ** 2
    return V, torch.Tensor(Vbins), bins


class Observable(torch.nn.Module):
    def __init__(self, system):
        super().__init__()
        check_system(system)
        self.device = system.device
        self.volume = system.get_volume()
        self.cell = torch.Tensor(system.get_cell()).diag().to(self.device)
        # replica-stacked systems: observables are per replica (frames = time x replica)
        self.n_rep = getattr(system, "n_replicas", 1)
        self.natoms = getattr(system, "group_size", system.get_number_of_atoms())


class rdf(Observable):
    def __init__(self, system, nbins, r_range, index_tuple=None, width=None):
        super().__init__(system)
        start, end = r_range[0], r_range[1]
        V, vol_bins, bins = generate_vol_bins(start, end, nbins, dim=system.dim)
        self.V = V
        self.vol_bins = vol_bins.to(self.device)
        self.r_axis = np.linspace(start, end, nbins)
        self.bins = bins
        # GaussianSmearing(start, stop=bins[-1], n_gaussians=nbins, width)   nff/nn/layers.py:54-61
        offsets = torch.linspace(start, float(bins[-1]), nbins)
        w = (offsets[1] - offsets[0]) if width is None else torch.tensor(float(width))
        self.register_buffer("offsets", offsets.to(self.device))
        self.spacing = float(offsets[1] - offsets[0]) if nbins > 1 else 0.0     # linspace: equally spaced
        self.width = float(w)
        self.coeff = float(-0.5 / torch.pow(w.to(torch.float32), 2))
        self.nbins = nbins
        self.cutoff_boundary = end + 5e-1
        self.index_tuple = index_tuple
        self._cell_struct = _lib.make_cell(self.cell)      # diagonal of the cell, as the reference
        self._mask = ops.build_mask(self.natoms, index_tuple, None, self.device)

    def forward(self, xyz):
        if self.n_rep > 1 and xyz.shape[-2] == self.n_rep * self.natoms:
            xyz = xyz.reshape(xyz.shape[:-2] + (self.n_rep, self.natoms, 3))
        count = ops.RdfRawFn.apply(xyz, self.offsets, self.coeff, self.cutoff_boundary,
                                   self._cell_struct, self._mask, self.spacing)
        norm = count.sum()
        count = count / norm
        rdf = count / (self.vol_bins / self.V)
        return count, self.bins, rdf


class vacf(Observable):
    def __init__(self, system, t_range):
        super().__init__(system)
        self.t_window = [i for i in range(1, t_range, 1)]

    def forward(self, vel):
        """Lags 0 .. t_range-1 of the velocity autocorrelation of vel [T, N, 3] (mean over frames, atoms and
        components per lag, torchmd/observable.py:158-163) from one fused HIP reduction (ops.VacfFn)."""
        return ops.VacfFn.apply(vel, len(self.t_window) + 1)
