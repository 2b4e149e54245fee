"""Correctly spelled alias of mdgrad_amd.sovlers (the reference module is torchmd/sovlers.py)."""
from .sovlers import *  # noqa: F401,F403
from .sovlers import (NHverlet_update, verlet_update, NHVerlet, Verlet, SOLVERS, odeint,  # noqa: F401
                      odeint_adjoint, OdeintAdjointMethod)
