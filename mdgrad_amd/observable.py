"""Observables with the reference's interface (torchmd/observable.py): generate_vol_bins
:10-21, Observable :24-31, rdf :33-76, vacf :153-163.  The pair search + Gaussian smearing +
histogram of rdf.forward is one HIP op (ops.RdfRawFn, csrc/rdf.hip); vacf is one fused reduction over the
velocity trajectory (ops.VacfFn, csrc/observe.hip)."""
import warnings

import numpy as np
import torch

from . import _lib, ops
from .system import check_system


def generate_vol_bins(start, end, nbins, dim):
    """(volume of the ball of radius `end`, volumes of the nbins shells between the equally spaced edges, the edges):
    the ideal-gas normalisation of g(r) (torchmd/observable.py:10-21).  Same constants in the same association as
    the reference, so the float32 results agree to the last bit."""
    if dim not in (2, 3):
        raise ValueError("generate_vol_bins: dim must be 2 or 3")
    edges = torch.linspace(start, end, nbins + 1)
    shell_coef, ball = (4 * np.pi / 3, (4 / 3) * np.pi * end ** 3) if dim == 3 else (np.pi, np.pi * end ** 2)
    shells = shell_coef * (edges[1:] ** dim - edges[:-1] ** dim)
    return ball, torch.Tensor(shells), edges


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
        self._warned_fused = False
        self.last_path = None           # "kernel" | "fused-trajectory": which path produced the last forward's histogram

    def _fused_raw(self, xyz):
        """The raw histogram of `xyz` if a fused trajectory launch already produced it (ops.fused_traj); otherwise
        None -- after registering this observable and the frame selection with the integrator, so that the NEXT
        launch produces it.  `xyz` must be the trajectory tensor itself or a slice of it along time that runs to
        the last frame (q_t, q_t[::k], q_t[s:], q_t[s::k])."""
        base = xyz if hasattr(xyz, "_mdg_traj") else getattr(xyz, "_base", None)
        tag = getattr(base, "_mdg_traj", None) if base is not None else None
        if tag is None or self._mask is not None or self.nbins < 2:
            return None
        spec, td, hint, raw = tag
        start, stride = 0, 1
        if xyz is not base:
            if (xyz.dim() != base.dim() or base.stride(td) == 0 or
                    any(xyz.shape[d] != base.shape[d] or xyz.stride(d) != base.stride(d) for d in range(base.dim()) if d != td)):
                return None
            off, bs = xyz.storage_offset() - base.storage_offset(), base.stride(td)
            if off < 0 or off % bs or xyz.stride(td) % bs or xyz.shape[td] < 1:
                return None
            start, stride = off // bs, max(1, xyz.stride(td) // bs)
            if start >= base.shape[td] or xyz.shape[td] != (base.shape[td] - start + stride - 1) // stride:
                return None
        if hint is not None and raw is not None and hint.matches(self, start, stride):
            if not self._warned_fused:
                self._warned_fused = True
                warnings.warn("mdgrad_amd.rdf: the histogram of this trajectory was produced inside the fused trajectory "
                              "launch; its dependence on the frames is routed through the adjoint launch (raw -> FusedTrajFn), "
                              "not through q_t in the autograd graph (autograd.grad(loss, q_t) / hooks on q_t do not see the "
                              "RDF term), and it is the fine-grid histogram (<= 2e-5 per bin from the exact kernel).  "
                              "(opted into with integrator.fuse_observables = True; rdf.last_path tells which path ran).",
                              stacklevel=3)
            return raw
        # OPT-IN (integrator.fuse_observables = True, or integrator.attach_observable): the fused histogram is an output of
        # the trajectory launch, not a function of q_t in the autograd graph -- a reference caller that differentiates
        # w.r.t. q_t or hooks it would lose the RDF term, so nothing is fused unless asked for
        integ = getattr(spec, "_integrator", None)
        if integ is not None and getattr(integ, "fuse_observables", False):
            new = ops.RdfFuse(self, start, stride)
            spec.rdf_hint = new
            integ._rdf_hint = new
        return None

    def forward(self, xyz):
        count = self._fused_raw(xyz)
        self.last_path = "fused-trajectory" if count is not None else "kernel"
        if count is None:
            if self.n_rep > 1 and xyz.shape[-2] == self.n_rep * self.natoms:
                xyz = xyz.reshape(xyz.shape[:-2] + (self.n_rep, self.natoms, 3))
            count = ops.RdfRawFn.apply(xyz, self.offsets, self.coeff, self.cutoff_boundary,
                                       self._cell_struct, self._mask, self.spacing, float(self.r_axis[-1]))
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
