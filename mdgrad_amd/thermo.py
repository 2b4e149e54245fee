"""torchmd/thermo.py:57-66 (`Temperature`); `Pressure` is dead code in the reference (undefined names)
and is not provided."""
import torch

from . import ops
from .observable import Observable


class Temperature(Observable):
    def __init__(self, system):
        super().__init__(system)
        self.dof = getattr(system, "dim", 3) * self.natoms              # N_dof = N * dim (thermo.py:63)
        self.mass = torch.Tensor(system.get_masses()).to(self.device)[:self.natoms].contiguous()

    def forward(self, velocities):
        """Instantaneous kinetic temperature (energy units), sum(m v^2) / N_dof: a scalar for one frame [N, 3] like
        the reference, one value per frame for a trajectory [T, N, 3] or a batched one [R, T, N, 3] (one fused launch,
        csrc/observe.hip)."""
        lead = velocities.shape[:-2]                      # any leading shape: [], [T], [R, T] (batched fused trajectories)
        n = velocities.shape[-2]
        if n % self.natoms:
            raise ValueError("Temperature: %d atoms per frame is not a multiple of the system's %d" % (n, self.natoms))
        k = n // self.natoms                              # replica-stacked state [..., R N, 3]: one value per replica
        v = velocities.reshape(-1, self.natoms, 3)
        out = ops.TemperatureFn.apply(v, self.mass, self.dof)
        return out.reshape(*lead, k) if k > 1 else out.reshape(lead)
