"""mdgrad_amd -- MI355X-native differentiable-MD hot path behind the torchmd/mdgrad API.

    from mdgrad_amd.system import System, FaceCenteredCubic
    from mdgrad_amd.potentials import LennardJones, ExcludedVolume
    from mdgrad_amd.interface import PairPotentials, Stack
    from mdgrad_amd.md import NoseHooverChain, NVE, Simulations
    from mdgrad_amd.observable import rdf

All numerics on the hot path run in libmdgrad_hip.so (hand-written HIP for gfx950, C ABI in
include/mdgrad_hip.h).  There is no CPU fallback: tensors must live on a HIP device.
"""
__version__ = "0.1.0"
