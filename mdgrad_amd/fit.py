"""Host-side pieces of the reference's RDF-fitting loop, with its function names:
get_exp_rdf (scripts/data.py:11-31), get_unit_len (:47-57, in units.py), JS_rdf (demo/fit_rdf_gnn.py:36-42),
compute_D (:410-411), get_temp (:117-118).  Plain torch / numpy; the simulation inside the loop is the
HIP hot path."""
import numpy as np
import torch

from .observable import generate_vol_bins
from .units import get_unit_len  # noqa: F401  (re-exported under the reference's name)


def get_exp_rdf(data, nbins, r_range, device, dim=3):
    """Target g(r) on the observable's grid from tabulated (r, g) data ([2, n] or [n, 2]): linear
    interpolation, then the same volume-weighted normalisation the rdf observable applies to its counts."""
    data = np.asarray(data, dtype=np.float64)
    r, g = (data[0], data[1]) if data.shape[0] == 2 else (data[:, 0], data[:, 1])
    start, end = r_range
    xnew = np.linspace(start, end, nbins)
    if xnew[0] < r.min() or xnew[-1] > r.max():
        raise ValueError("get_exp_rdf: r_range outside the tabulated data")     # interp1d raises here too
    V, vol_bins, _ = generate_vol_bins(start, end, nbins, dim=dim)
    vol_bins = vol_bins.to(device)
    g_obs = torch.Tensor(np.interp(xnew, r, g)).to(device)
    g_obs = g_obs * (V / float((g_obs * vol_bins).sum()))
    return xnew, g_obs


def JS_rdf(g_obs, g, e0=1e-4):
    """Jensen-Shannon-style divergence between two RDFs."""
    g_m = 0.5 * (g_obs + g)
    js = (-(g_obs + e0) * (torch.log(g_m + e0) - torch.log(g_obs + e0))).mean()
    return js + (-(g + e0) * (torch.log(g_m + e0) - torch.log(g + e0))).mean()


def compute_D(dev, rho, rrange):
    """Volume-weighted squared RDF deviation, sum 4 pi rho r^2 dev^2 dr."""
    return (4 * np.pi * rho * (rrange ** 2) * dev ** 2 * (rrange[2] - rrange[1])).sum()


def get_temp(T_start, T_equil, n_epochs, i, anneal_rate):
    """Exponential annealing schedule from T_start to T_equil."""
    return (T_start - T_equil) * np.exp(-i * (1 / n_epochs) * anneal_rate) + T_equil
