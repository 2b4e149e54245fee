"""torchmd/thermo.py:57-66 (`Temperature`); `Pressure` is dead code in the reference (undefined names)
and is not provided."""
import torch

from .observable import Observable


class Temperature(Observable):
    def __init__(self, system):
        super().__init__(system)
        self.dof = 3 * self.natoms
        self.mass = torch.Tensor(system.get_masses()).to(self.device)

    def forward(self, velocities):
        """Instantaneous kinetic temperature (energy units) of each frame: sum(m v^2) / dof."""
        return (self.mass[:, None] * velocities.pow(2)).sum((-1, -2)) / self.dof
