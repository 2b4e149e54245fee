"""SchNet as the reference evaluates it under GNNPotentials.

The irregular part of every interaction block -- gather both endpoints, multiply by the
continuous filter, scatter-add both directions (nff/nn/modules.py:564-571,
nff/nn/graphconv.py:43-53) -- is ONE HIP gather (ops.CfconvAggFn, csrc/graph.hip) over the
per-atom list when the batch carries the topology built by GNNPotentials; its derivatives are
the same kernel family, so forces (first order) and the adjoint's force-vjp (second order)
stay on HIP kernels.  Dense layers are plain library GEMMs.  Distances use the raw image flags
like the reference (schnet.py:140-142) unless `cartesian_offsets=True`.
"""
import collections

import numpy as np
import torch
import torch.nn as nn
from torch.nn import ModuleDict, Sequential

from .. import ops
from .graphconv import MessagePassingModule
from .graphop import batch_and_sum
from .layers import Dense, GaussianSmearing, shifted_softplus

_LAYER_TYPES = {"linear": torch.nn.Linear, "Tanh": torch.nn.Tanh, "ReLU": torch.nn.ReLU, "Dense": Dense,
                "shifted_softplus": shifted_softplus}


def get_default_readout(n_atom_basis):                       # nff/nn/utils.py:56-75
    return {'energy': [
        {'name': 'linear', 'param': {'in_features': n_atom_basis, 'out_features': int(n_atom_basis / 2)}},
        {'name': 'shifted_softplus', 'param': {}},
        {'name': 'linear', 'param': {'in_features': int(n_atom_basis / 2), 'out_features': 1}}]}


def construct_sequential(layers):                            # nff/nn/utils.py:22-38
    return Sequential(collections.OrderedDict(
        [layer['name'] + str(i), _LAYER_TYPES[layer['name']](**layer['param'])] for i, layer in enumerate(layers)))


class NodeMultiTaskReadOut(nn.Module):                       # nff/nn/modules.py:761-809
    def __init__(self, multitaskdict, post_readout=None):
        super().__init__()
        self.readout = ModuleDict({k: construct_sequential(v) for k, v in multitaskdict.items()})
        self.post_readout = post_readout
        self.multitaskdict = multitaskdict

    def forward(self, r):
        out = {key: self.readout[key](r) for key in self.readout}
        if self.post_readout is not None:
            out = self.post_readout(out, self.multitaskdict)
        return out


class SchNetConv(MessagePassingModule):
    """nff/nn/modules.py:514-575 on MessagePassingModule (graphconv.py:11-53): `message` = filter network x filtered node rows
    at both ends of a pair, `aggregate` = scatter-add, `update` = the update MLP.  With the topology GNNPotentials attaches,
    `forward` replaces message + aggregate by one atom-centric HIP gather."""

    def __init__(self, n_atom_basis, n_filters, n_gaussians, cutoff, trainable_gauss):
        super().__init__()
        self.moduledict = ModuleDict({
            'message_edge_filter': Sequential(
                GaussianSmearing(start=0.0, stop=cutoff, n_gaussians=n_gaussians, trainable=trainable_gauss),
                Dense(in_features=n_gaussians, out_features=n_gaussians),
                shifted_softplus(),
                Dense(in_features=n_gaussians, out_features=n_filters)),
            'message_node_filter': Dense(in_features=n_atom_basis, out_features=n_filters),
            'update_function': Sequential(
                Dense(in_features=n_filters, out_features=n_atom_basis),
                shifted_softplus(),
                Dense(in_features=n_atom_basis, out_features=n_atom_basis)),
        })

    fused_filter = True          # K9 on the matrix cores (csrc/cfconv_filter.hip)
    filter_bf16 = False          # bf16 MFMA operands (fp32 accumulate); default fp32 MFMA = exact f32

    def edge_filter(self, e):
        seq = self.moduledict['message_edge_filter']
        smear, d1, d2 = seq[0], seq[1], seq[3]
        if (self.fused_filter and e.is_cuda and not smear.centered and smear.offsets.shape[0] <= 64
                and d1.activation is None and d2.activation is None and d1.bias is not None
                and d2.bias is not None):
            return ops.CfconvFilterFn.apply(e.reshape(-1), smear.offsets, smear.width, d1.weight, d1.bias,
                                            d2.weight, d2.bias, self.filter_bf16)
        return seq(e)

    def message(self, r, e, a, aggr_wgt=None):               # modules.py:547-568
        W = self.edge_filter(e)                              # [E,F] continuous filter
        h = self.moduledict['message_node_filter'](r)        # [N,F]
        if aggr_wgt is not None:
            h = h * aggr_wgt
        return h[a[:, 0]] * W, h[a[:, 1]] * W

    def update(self, r):                                     # modules.py:570-571
        return self.moduledict['update_function'](r)

    def forward(self, r, e, a, aggr_wgt=None, topo=None):
        if topo is None:                                     # explicit list without topology: the reference's own three steps
            return MessagePassingModule.forward(self, r, e, a, aggr_wgt)
        W = self.edge_filter(e)
        h = self.moduledict['message_node_filter'](r)
        if aggr_wgt is not None:
            h = h * aggr_wgt
        return self.update(ops.CfconvAggFn.apply(h, W, topo))


class SchNet(nn.Module):
    """nff/nn/models/schnet.py:23-171."""

    # Analytic path (mdgrad_amd/nn/analytic.py), with filter_bf16 on the row-chain kernels: the node matrices the convolution
    # kernels GATHER per edge are read from bf16 mirrors (mdg_cfconv_*_rows16, include/mdgrad_hip.h).  A precision option of
    # its own on top of the bf16 MFMA operands -- the gathered features carry 8 significant bits into f32 products -- and off
    # unless asked for; tests/test_gpu_schnet_rows16.py holds its tolerance.
    node_rows_bf16 = False
    # With node_rows_bf16 (every block gathering bf16 rows): the node-level Dense chains form each product as three bf16 MFMAs on
    # operands split into a bf16 head and remainder (MDG_CHAIN_X3, csrc/rowchain.hip: ~1e-5 per product, f32 accumulate).
    # False: exact f32 products there.
    chain_x3 = True
    # True (or MDG_CHAIN_X6=1): with f32 rows the chains use six exact-bf16-piece products per Dense product (MDG_CHAIN_X6: f32
    # accuracy on the bf16 matrix pipe).  Off by default: 1-4 % slower in the pass than v_mfma_f32_16x16x4_f32
    # (profiles/r06_chain_x6_ab.txt).
    chain_x6 = False
    # One C-ABI call per force / force-vjp evaluation (nn/plan.py, csrc/schnet_eval.hip) and, inside it, the per-edge stash of
    # the filter network's first layer for the rows16 sweeps -- same kernels and results as the launch-by-launch path / the
    # recomputing sweeps; False selects those (A/B and debugging).
    eval_plan = True
    filter_stash = True

    def __init__(self, modelparams):
        super().__init__()
        n_atom_basis = modelparams['n_atom_basis']
        n_filters = modelparams['n_filters']
        n_gaussians = modelparams['n_gaussians']
        n_convolutions = modelparams['n_convolutions']
        cutoff = modelparams['cutoff']
        trainable_gauss = modelparams.get('trainable_gauss', False)
        self.cartesian_offsets = modelparams.get('cartesian_offsets', False)
        self.atom_embed = nn.Embedding(100, n_atom_basis, padding_idx=0)
        readoutdict = modelparams.get('readoutdict', get_default_readout(n_atom_basis))
        post_readout = modelparams.get('post_readout', None)
        self.convolutions = nn.ModuleList([
            SchNetConv(n_atom_basis=n_atom_basis, n_filters=n_filters, n_gaussians=n_gaussians, cutoff=cutoff,
                       trainable_gauss=trainable_gauss) for _ in range(n_convolutions)])
        self.atomwisereadout = NodeMultiTaskReadOut(multitaskdict=readoutdict, post_readout=post_readout)
        self.device = None

    def convolve(self, batch, xyz=None):
        if xyz is None:
            xyz = batch['nxyz'][:, 1:4]
            xyz.requires_grad = True
        r = batch['nxyz'][:, 0]
        N = batch['num_atoms'].reshape(-1).tolist()
        a = batch['nbr_list']
        offsets = batch.get('offsets', 0)
        topo = batch.get('_topo', None)
        if self.cartesian_offsets and torch.is_tensor(offsets) and 'cell' in batch:
            offsets = offsets.matmul(batch['cell'])
        if topo is not None:
            dvec = ops.EdgeDiffFn.apply(xyz, topo) - offsets
        else:
            dvec = xyz[a[:, 0]] - xyz[a[:, 1]] - offsets
        e = dvec.pow(2).sum(1).sqrt()[:, None]               # schnet.py:142
        r = self.atom_embed(r.long()).squeeze()
        for conv in self.convolutions:
            r = r + conv(r=r, e=e, a=a, topo=topo)            # schnet.py:149-151
        return r, N, xyz

    def forward(self, batch, xyz=None):
        r, N, xyz = self.convolve(batch, xyz)
        r = self.atomwisereadout(r)
        return batch_and_sum(r, N, list(batch.keys()), xyz)  # schnet.py:167-169


_PARAMS_TYPE = {'n_atom_basis': int, 'n_filters': int, 'n_gaussians': int, 'n_convolutions': int,
                'cutoff': float, 'bond_par': float, 'trainable_gauss': bool, 'box_size': np.ndarray}


class ParameterError(Exception):
    pass


def get_model(params, model_type="SchNet", **kwargs):
    """nff/train/builders/model.py:92-106 (type-checks the hyper-parameters, SchNet only)."""
    if model_type != "SchNet":
        raise NotImplementedError("mdgrad_amd.nn provides SchNet only")
    for key, val in params.items():
        if key in _PARAMS_TYPE and not isinstance(val, _PARAMS_TYPE[key]):
            raise ParameterError('%s is not %s' % (str(key), _PARAMS_TYPE[key]))
    return SchNet(params, **kwargs)
