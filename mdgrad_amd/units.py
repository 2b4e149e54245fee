"""Unit constants the reference takes from `ase.units` (ase 3.20.1, CODATA 2014;
call sites torchmd/md.py:5,73, torchmd/interface.py:8).  ase is not a dependency."""
kB = 8.617330337217213e-05      # eV / K
fs = 0.09822694788464063        # Angstrom * sqrt(amu / eV)
C = 6.241509125883258e+18
m = 1e10
_Nav = 6.022140857e+23


def get_unit_len(rho, mass, N_unitcell):
    """Lattice constant (Angstrom) for density rho [g/cm^3], molar mass [g/mol] and
    N_unitcell particles per cubic cell (scripts/data.py:47-57)."""
    Na = 6.02214086e+23
    N = (rho * 10 ** 6 / mass) * Na
    rho_n = N / (10 ** 30)
    return (N_unitcell / rho_n) ** (1 / 3)
