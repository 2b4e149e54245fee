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
        the reference, one value per frame for a trajectory [T, N, 3] (one fused launch, csrc/observe.hip)."""
        v = velocities if velocities.dim() == 3 else velocities[None]
        T_, n = v.shape[0], v.shape[1]
        if n != self.natoms:                     # replica-stacked state [T, R N, 3]: one temperature per (frame, replica)
            out = ops.TemperatureFn.apply(v.reshape(T_ * (n // self.natoms), self.natoms, 3), self.mass, self.dof)
            out = out.reshape(T_, n // self.natoms)
        else:
            out = ops.TemperatureFn.apply(v, self.mass, self.dof)
        return out if velocities.dim() == 3 else out[0]
