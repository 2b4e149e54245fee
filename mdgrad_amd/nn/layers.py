"""Radial basis, dense layer and activation used by the SchNet blocks.  Public names, constructor
arguments and state_dict keys follow nff/nn/layers.py:14-134 and nff/nn/activations.py:5-11 so that
reference checkpoints load unchanged; the code is this package's own."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

_LN2 = math.log(2.0)


def gaussian_smearing(distances, offset, widths, centered=False):
    """Gaussian radial basis.  Regular mode: exp(-(d - mu_k)^2 / (2 w_k^2)); centred mode: the offsets
    act as widths of origin-centred Gaussians, exp(-d^2 / (2 mu_k^2))  (layers.py:14-31)."""
    sigma, shift = (offset, 0.0) if centered else (widths, offset)
    return torch.exp((distances - shift).pow(2) * (-0.5 / sigma.pow(2)))


class GaussianSmearing(nn.Module):
    """n_gaussians centres on linspace(start, stop); one shared width (centre spacing unless `width`
    is given).  `width` / `offsets` are buffers, or Parameters when trainable (layers.py:34-83)."""

    def __init__(self, start, stop, n_gaussians, width=None, centered=False, trainable=False):
        super().__init__()
        centres = torch.linspace(start, stop, n_gaussians)
        w = (centres[1] - centres[0]) if width is None else width
        widths = torch.ones_like(centres) * w
        self.centered = centered
        for name, value in (("width", widths), ("offsets", centres)):
            if trainable:
                setattr(self, name, nn.Parameter(value))
            else:
                self.register_buffer(name, value)

    def forward(self, distances):
        return gaussian_smearing(distances, self.offsets, self.width, centered=self.centered)


class Dense(nn.Linear):
    """Linear layer with optional activation, Xavier-uniform weight and zero bias by default
    (layers.py:86-134).  2-D f32 HIP inputs run on the hand-written kernels -- the product on the MFMA node kernel
    (mdg_dense), the weight gradient on the split-K kernel -- closed under differentiation (ops.linear); CPU / other inputs
    take torch's own."""

    def __init__(self, in_features, out_features, bias=True, activation=None,
                 weight_init=nn.init.xavier_uniform_, bias_init=None):
        self.weight_init = weight_init
        self.bias_init = nn.init.zeros_ if bias_init is None else bias_init
        self.activation = activation
        super().__init__(in_features, out_features, bias)

    def reset_parameters(self):
        self.weight_init(self.weight)
        if self.bias is not None:
            self.bias_init(self.bias)

    def forward(self, inputs):
        if inputs.is_cuda and inputs.dim() == 2 and inputs.dtype == torch.float32:
            from .. import ops
            y = ops.linear(inputs, self.weight, self.bias)       # (MFMA node kernel + split-K weight gradient: no library GEMM)
        else:
            y = F.linear(inputs, self.weight, self.bias)
        return self.activation(y) if self.activation else y


class shifted_softplus(nn.Module):
    """softplus(x) - ln 2."""

    def forward(self, input):
        return F.softplus(input) - _LN2
