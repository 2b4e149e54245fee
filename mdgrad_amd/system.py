"""State container: `System` with the reference's interface (torchmd/system.py:16-70) but
without ase -- a minimal Atoms-like base carries positions / numbers / cell / masses /
momenta on the host in float64, exactly the state ase.Atoms holds for the reference.
An object exposing the ase.Atoms getters (including a real ase.Atoms) can be passed as the
first argument, mirroring `System(atoms, device=...)`.
"""
import numpy as np
import torch

from . import units

# standard atomic weights for the elements the demos use (ase.data.atomic_masses values)
_MASSES = {1: 1.008, 2: 4.002602, 3: 6.94, 4: 9.0121831, 5: 10.81, 6: 12.011, 7: 14.007, 8: 15.999,
           9: 18.998403163, 10: 20.1797, 11: 22.98976928, 12: 24.305, 13: 26.9815385, 14: 28.085,
           15: 30.973761998, 16: 32.06, 17: 35.45, 18: 39.948, 19: 39.0983, 20: 40.078,
           26: 55.845, 29: 63.546, 36: 83.798, 54: 131.293}
_SYMBOLS = {"H": 1, "He": 2, "Li": 3, "Be": 4, "B": 5, "C": 6, "N": 7, "O": 8, "F": 9, "Ne": 10, "Na": 11,
            "Mg": 12, "Al": 13, "Si": 14, "P": 15, "S": 16, "Cl": 17, "Ar": 18, "K": 19, "Ca": 20,
            "Fe": 26, "Cu": 29, "Kr": 36, "Xe": 54}


def wrap_positions(positions, cell, pbc=True, center=(0.5, 0.5, 0.5), eps=1e-7):
    """Fractional coordinates mod 1 with ase's eps shift (ase.geometry.wrap_positions, used at
    torchmd/md.py:66 and by Atoms.get_positions(wrap=True))."""
    cell = np.asarray(cell, dtype=np.float64)
    if cell.ndim == 1:
        cell = np.diag(cell)
    shift = np.asarray(center, dtype=np.float64) - 0.5 - eps
    pos = np.asarray(positions, dtype=np.float64)
    inv = np.linalg.inv(cell)
    # explicit 3-term sums instead of [N,3]@[3,3]: a threaded BLAS call here wakes one spinning
    # worker per host core, which is enough to get a CPU-quota'd container throttled for tens
    # of ms (measured on the MI355X box: forward 105 ms -> 15 ms, see DESIGN.md "host side")
    if not (cell - np.diag(np.diag(cell))).any():
        # orthorhombic cell: the off-diagonal terms of the sums below are +0.0, so the three per-axis products give the same
        # bits with a third of the arithmetic (this runs inside every pass of a training loop)
        d = np.diag(cell)
        frac = pos * np.diag(inv) - shift
        frac -= np.floor(frac)               # = frac % 1.0 bit for bit (the subtraction is exact), five times faster in numpy
        frac += shift
        frac *= d
        return frac
    frac = pos[:, 0:1] * inv[0] + pos[:, 1:2] * inv[1] + pos[:, 2:3] * inv[2] - shift
    frac %= 1.0
    frac += shift
    return frac[:, 0:1] * cell[0] + frac[:, 1:2] * cell[1] + frac[:, 2:3] * cell[2]


class Atoms:
    """The subset of ase.Atoms the hot path touches."""

    def __init__(self, symbols=None, positions=None, numbers=None, cell=None, pbc=True, masses=None,
                 momenta=None, velocities=None):
        if symbols is not None and hasattr(symbols, "get_positions"):
            src = symbols
            positions = src.get_positions()
            numbers = src.get_atomic_numbers()
            cell = np.asarray(src.get_cell())
            masses = src.get_masses()
            momenta = src.get_momenta() if hasattr(src, "get_momenta") else None
            pbc = getattr(src, "pbc", pbc)
        elif symbols is not None and numbers is None:
            if isinstance(symbols, str):
                symbols = [symbols] * len(positions)
            numbers = [_SYMBOLS[s] for s in symbols]
        self.positions = np.array(positions, dtype=np.float64).reshape(-1, 3)
        n = len(self.positions)
        self.numbers = (np.array(numbers, dtype=np.int64).reshape(-1) if numbers is not None
                        else np.ones(n, dtype=np.int64))
        c = np.zeros((3, 3)) if cell is None else np.array(cell, dtype=np.float64)
        self.cell = np.diag(c) if c.ndim == 1 else c
        if masses is None:
            masses = [_MASSES.get(int(z), 1.0) for z in self.numbers]
        self.masses = np.array(masses, dtype=np.float64).reshape(-1)
        self.momenta = np.zeros((n, 3))
        if momenta is not None:
            self.momenta = np.array(momenta, dtype=np.float64).reshape(-1, 3)
        if velocities is not None:
            self.set_velocities(velocities)
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
            return wrap_positions(self.positions, self.cell, **kw)
        return self.positions.copy()

    def set_positions(self, p):
        self.positions = np.array(p, dtype=np.float64).reshape(-1, 3)

    def get_cell(self):
        return self.cell.copy()

    def set_cell(self, cell):
        c = np.array(cell, dtype=np.float64)
        self.cell = np.diag(c) if c.ndim == 1 else c

    def get_volume(self):
        return abs(float(np.linalg.det(self.cell)))

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

    def get_kinetic_energy(self):
        return 0.5 * float((self.momenta ** 2 / self.masses[:, None]).sum())

    def get_temperature(self):
        return 2 * self.get_kinetic_energy() / (3 * len(self) * units.kB)


def _cubic_lattice(basis, symbol, size, latticeconstant, pbc=True):
    if isinstance(size, int):
        size = (size, size, size)
    a = float(latticeconstant)
    pts = [(np.array([i, j, k], dtype=np.float64) + b) * a
           for i in range(size[0]) for j in range(size[1]) for k in range(size[2]) for b in basis]
    return Atoms(symbols=[symbol] * len(pts), positions=np.array(pts),
                 cell=np.array([a * size[0], a * size[1], a * size[2]]), pbc=pbc)


_FCC = np.array([[0, 0, 0], [.5, .5, 0], [.5, 0, .5], [0, .5, .5]], dtype=np.float64)
_DIA = np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0], [.25, .25, .25], [.25, .75, .75],
                 [.75, .25, .75], [.75, .75, .25]], dtype=np.float64)


def FaceCenteredCubic(symbol="H", size=(1, 1, 1), latticeconstant=1.0, pbc=True, **_):
    """Cubic FCC supercell, 4*nx*ny*nz atoms (stand-in for ase.lattice.cubic.FaceCenteredCubic)."""
    return _cubic_lattice(_FCC, symbol, size, latticeconstant, pbc)


def Diamond(symbol="H", size=(1, 1, 1), latticeconstant=1.0, pbc=True, **_):
    """Cubic diamond supercell, 8*nx*ny*nz atoms (stand-in for ase.lattice.cubic.Diamond)."""
    return _cubic_lattice(_DIA, symbol, size, latticeconstant, pbc)


def check_system(obj):
    """torchmd/system.py:11-14."""
    if obj.__class__ != System:
        raise TypeError("input should be a mdgrad_amd.system.System")


class System(Atoms):
    """torchmd/system.py:16-70.  `device` is a torch device ("cuda:0" / int index)."""

    def __init__(self, *args, device, dim=3, props=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.props = {} if props is None else props
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.dim = dim
        self.n_replicas = 1
        self.group_size = len(self)

    def replicate(self, n_replicas):
        """Stack n independent replicas of this system into one System of n*N atoms (extension; the
        reference walks a Python list of simulations, demo/fit_rdf_gnn.py:386-399).  Atoms of
        different replicas never interact; every replica gets its own thermostat chain.  Positions /
        velocities of the copies start identical -- perturb them with set_positions / set_velocities
        (shape [n*N, 3], replica-major)."""
        N = len(self)
        rep = System(positions=np.tile(self.positions, (n_replicas, 1)), numbers=np.tile(self.numbers, n_replicas),
                     cell=self.cell, masses=np.tile(self.masses, n_replicas),
                     momenta=np.tile(self.momenta, (n_replicas, 1)), pbc=self.pbc, device=self.device,
                     dim=self.dim, props=self.props)
        rep.n_replicas, rep.group_size = n_replicas, N
        return rep

    def get_nxyz(self):                                      # system.py:39-51
        return np.concatenate([self.get_atomic_numbers().reshape(-1, 1),
                               self.get_positions().reshape(-1, 3)], axis=1)

    def get_cell_len(self):                                  # system.py:53-54
        return np.diag(self.get_cell())

    def get_batch(self):                                     # system.py:56-62
        return {"nxyz": torch.Tensor(self.get_nxyz()),
                "num_atoms": torch.LongTensor([self.group_size] * self.n_replicas),
                "energy": 0.0}

    def set_temperature(self, T, rng=None):                  # system.py:64-70
        """Maxwell-Boltzmann momenta xi*sqrt(m kT), T in energy units (as
        ase.md.velocitydistribution.MaxwellBoltzmannDistribution(atoms, T) in ase 3.20)."""
        rng = np.random.default_rng() if rng is None else rng
        m = self.get_masses()
        self.set_momenta(rng.standard_normal((len(m), 3)) * np.sqrt(m * T)[:, None])
        if self.dim < 3:
            vel = self.get_velocities()
            vel[:, -1] = 0.0
            self.set_velocities(vel)
