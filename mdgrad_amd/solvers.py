"""Correctly spelled alias of `mdgrad_amd.sovlers` (the reference's module is torchmd/sovlers.py; SURVEY §7 lists
`solvers.py`): `import mdgrad_amd.solvers` and `from mdgrad_amd.solvers import odeint_adjoint` both work."""
from .sovlers import *                                                                    # noqa: F401,F403
from .sovlers import NHverlet_update, verlet_update, NHVerlet, Verlet, odeint, odeint_adjoint, OdeintAdjointMethod  # noqa: F401
