"""SchNet pieces with the reference's class names and state_dict layout (nff/nn):
GaussianSmearing / Dense (layers.py), shifted_softplus (activations.py), SchNetConv
(modules.py:514-575), NodeMultiTaskReadOut (modules.py:761-809), SchNet (models/schnet.py:23-171),
get_model (nff/train/builders/model.py:92-106), MessagePassingModule (graphconv.py:11-53), scatter_add / compute_grad
(nff/utils/scatter.py), split_and_sum / batch_and_sum (graphop.py:9-63)."""
from .layers import GaussianSmearing, Dense, shifted_softplus, gaussian_smearing          # noqa: F401
from .schnet import SchNetConv, NodeMultiTaskReadOut, SchNet, get_model, get_default_readout  # noqa: F401
from .graphconv import MessagePassingModule                                                # noqa: F401
from .graphop import scatter_add, compute_grad, split_and_sum, batch_and_sum              # noqa: F401
