"""nff/nn/layers.py:14-134 and nff/nn/activations.py:5-11 (same names, same state_dict keys)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.init import xavier_uniform_, constant_


def gaussian_smearing(distances, offset, widths, centered=False):
    """exp(-0.5/width^2 (d - mu)^2)  (layers.py:14-31)."""
    if not centered:
        coeff = -0.5 / torch.pow(widths, 2)
        diff = distances - offset
    else:
        coeff = -0.5 / torch.pow(offset, 2)
        diff = distances
    return torch.exp(coeff * torch.pow(diff, 2))


class GaussianSmearing(nn.Module):
    """layers.py:34-83: offsets = linspace(start, stop, n), width = offsets[1]-offsets[0]
    unless given; buffers unless trainable."""

    def __init__(self, start, stop, n_gaussians, width=None, centered=False, trainable=False):
        super().__init__()
        offset = torch.linspace(start, stop, n_gaussians)
        if width is None:
            widths = torch.FloatTensor((offset[1] - offset[0]) * torch.ones_like(offset))
        else:
            widths = torch.FloatTensor(width * torch.ones_like(offset))
        if trainable:
            self.width = nn.Parameter(widths)
            self.offsets = nn.Parameter(offset)
        else:
            self.register_buffer('width', widths)
            self.register_buffer('offsets', offset)
        self.centered = centered

    def forward(self, distances):
        return gaussian_smearing(distances, self.offsets, self.width, centered=self.centered)


class Dense(nn.Linear):
    """y = activation(x W^T + b), xavier-uniform W, zero b (layers.py:86-134).  The GEMM is the
    library one (rocBLAS/hipBLASLt through torch)."""

    def __init__(self, in_features, out_features, bias=True, activation=None,
                 weight_init=xavier_uniform_, bias_init=None):
        self.weight_init = weight_init
        self.bias_init = bias_init if bias_init is not None else (lambda b: constant_(b, 0.))
        self.activation = activation
        super().__init__(in_features, out_features, bias)

    def reset_parameters(self):
        self.weight_init(self.weight)
        if self.bias is not None:
            self.bias_init(self.bias)

    def forward(self, inputs):
        if inputs.is_cuda and inputs.dim() == 2 and inputs.dtype == torch.float32 and inputs.shape[0] >= 8192:
            from .. import ops
            y = ops.linear(inputs, self.weight, self.bias)      # weight gradient on the split-K MFMA kernel
        else:
            y = super().forward(inputs)
        if self.activation:
            y = self.activation(y)
        return y


class shifted_softplus(nn.Module):
    """softplus(x) - ln 2  (activations.py:5-11)."""

    def forward(self, input):
        return F.softplus(input) - math.log(2.0)
