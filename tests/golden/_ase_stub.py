"""Minimal in-memory stand-ins for `ase` and `xitorch`, used ONLY by
tests/golden/make_goldens.py in the build container so that the reference at
/root/reference can be imported and run to produce golden vectors.

This is our own code (written from the public ase API, not copied); it never
travels as part of the product and nothing under mdgrad_amd/ imports it.
Only the handful of entry points the reference's hot path touches exist here
(SURVEY.md 8c): ase.Atoms, ase.units, ase.geometry.wrap_positions,
ase.md.velocitydistribution.MaxwellBoltzmannDistribution, ase.lattice.cubic,
xitorch.interpolate.Interp1D.
"""
import sys
import types

import numpy as np


class Atoms:
    def __init__(self, symbols=None, positions=None, numbers=None, cell=None,
                 pbc=True, masses=None, momenta=None, **_):
        if isinstance(symbols, Atoms):
            src = symbols
            positions = src.positions.copy()
            numbers = src.numbers.copy()
            cell = src.cell.copy()
            masses = src.masses.copy()
            momenta = src.momenta.copy()
            pbc = src.pbc
        self.positions = np.array(positions, dtype=np.float64).reshape(-1, 3)
        n = len(self.positions)
        self.numbers = (np.array(numbers, dtype=np.int64) if numbers is not None
                        else np.ones(n, dtype=np.int64))
        c = np.array(cell, dtype=np.float64)
        self.cell = np.diag(c) if c.ndim == 1 else c
        self.masses = (np.array(masses, dtype=np.float64) if masses is not None
                       else np.full(n, 1.008))
        self.momenta = (np.array(momenta, dtype=np.float64).reshape(-1, 3)
                        if momenta is not None else np.zeros((n, 3)))
        self.pbc = pbc

    def __len__(self):
        return len(self.positions)

    def get_number_of_atoms(self):
        return len(self.positions)

    get_global_number_of_atoms = get_number_of_atoms

    def get_atomic_numbers(self):
        return self.numbers.copy()

    def get_positions(self, wrap=False, **kw):
        if wrap:
            return wrap_positions(self.positions, self.cell)
        return self.positions.copy()

    def set_positions(self, p):
        self.positions = np.array(p, dtype=np.float64).reshape(-1, 3)

    def get_cell(self):
        return self.cell.copy()

    def get_volume(self):
        return abs(np.linalg.det(self.cell))

    def get_masses(self):
        return self.masses.copy()

    def get_momenta(self):
        return self.momenta.copy()

    def set_momenta(self, m):
        self.momenta = np.array(m, dtype=np.float64).reshape(-1, 3)

    def get_velocities(self):
        return self.momenta / self.masses[:, None]

    def set_velocities(self, v):
        self.momenta = np.array(v, dtype=np.float64).reshape(-1, 3) * self.masses[:, None]


def wrap_positions(positions, cell, pbc=True, center=(0.5, 0.5, 0.5),
                   pretty_translation=False, eps=1e-7):
    cell = np.asarray(cell, dtype=np.float64)
    shift = np.asarray(center, dtype=np.float64) - 0.5 - eps
    frac = np.linalg.solve(cell.T, np.asarray(positions, dtype=np.float64).T).T - shift
    frac %= 1.0
    frac += shift
    return frac @ cell


def MaxwellBoltzmannDistribution(atoms, temp=None, temperature_K=None, rng=None, **_):
    rng = np.random if rng is None else rng
    kT = temp if temp is not None else temperature_K * 8.617330337217213e-05
    m = atoms.get_masses()
    xi = rng.standard_normal((len(m), 3))
    atoms.set_momenta(xi * np.sqrt(m * kT)[:, None])


def install():
    ase = types.ModuleType("ase")
    ase.Atoms = Atoms
    units = types.ModuleType("ase.units")
    units.kB = 8.617330337217213e-05
    units.fs = 0.09822694788464063
    units.C = 6.241509125883258e+18
    units.m = 1e10
    ase.units = units
    geometry = types.ModuleType("ase.geometry")
    geometry.wrap_positions = wrap_positions
    ase.geometry = geometry
    md = types.ModuleType("ase.md")
    vd = types.ModuleType("ase.md.velocitydistribution")
    vd.MaxwellBoltzmannDistribution = MaxwellBoltzmannDistribution
    md.velocitydistribution = vd
    ase.md = md
    lattice = types.ModuleType("ase.lattice")
    cubic = types.ModuleType("ase.lattice.cubic")
    lattice.cubic = cubic
    ase.lattice = lattice
    xitorch = types.ModuleType("xitorch")
    interp = types.ModuleType("xitorch.interpolate")
    interp.Interp1D = object
    xitorch.interpolate = interp
    for name, mod in [("ase", ase), ("ase.units", units), ("ase.geometry", geometry),
                      ("ase.md", md), ("ase.md.velocitydistribution", vd),
                      ("ase.lattice", lattice), ("ase.lattice.cubic", cubic),
                      ("xitorch", xitorch), ("xitorch.interpolate", interp)]:
        sys.modules[name] = mod
