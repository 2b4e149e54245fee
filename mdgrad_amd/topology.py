"""Topology helpers with the reference's names (torchmd/topology.py).  The neighbour search
runs in the HIP builders (csrc/nbr.hip); results come back in the reference's format."""
import itertools

import torch

from . import _lib, ops


def compute_dis(xyz, nbr_list, offsets, cell):
    """torchmd/topology.py:5-12 (torch ops on the device; the pair kernels fuse this)."""
    nbr_list = nbr_list.to(xyz.device)
    cell = torch.diag(cell) if cell.dim() == 1 else cell
    return (xyz[nbr_list[:, 0]] - xyz[nbr_list[:, 1]] - offsets.matmul(cell)).pow(2).sum(1).sqrt()[:, None]


def generate_pair_index(N, index_tuple):
    """torchmd/topology.py:15-27."""
    mask_sel = torch.zeros(N, N)
    if index_tuple is not None:
        pair_mask = torch.LongTensor([list(items) for items in itertools.product(index_tuple[0], index_tuple[1])])
        mask_sel[pair_mask[:, 0], pair_mask[:, 1]] = 1
        mask_sel[pair_mask[:, 1], pair_mask[:, 0]] = 1
    return mask_sel


def generate_nbr_list(xyz, cutoff, cell, index_tuple=None, ex_pairs=None, get_dis=False):
    """torchmd/topology.py:30-73: minimum-image half list (i<j, lexicographic; a leading frame
    column for batched input) and image offsets.  xyz must be a HIP tensor."""
    _lib.require_gpu(xyz, "xyz")
    cs = _lib.make_cell(cell)
    N = xyz.shape[-2]
    mask = ops.build_mask(N, index_tuple, ex_pairs, xyz.device)
    frames = xyz.reshape(-1, N, 3)
    cellm = torch.as_tensor(cell, dtype=torch.float32, device=xyz.device)
    cellm = torch.diag(cellm) if cellm.dim() == 1 else cellm
    nbrs, offs, diss = [], [], []
    for f in range(frames.shape[0]):
        ell = ops.build_ell(frames[f], cs, cutoff, mask)
        nbr, off = ell.half_list()
        if xyz.dim() > 2:
            nbr = torch.cat([torch.full((nbr.shape[0], 1), f, dtype=nbr.dtype, device=nbr.device), nbr], 1)
        nbrs.append(nbr)
        offs.append(off)
        if get_dis:
            diss.append(compute_dis(frames[f], nbr[:, -2:], off, cellm).reshape(-1))
    nbr, off = torch.cat(nbrs), torch.cat(offs)
    if get_dis:
        return nbr, torch.cat(diss), off
    return nbr, off


def get_offsets(vecs, cell, device):
    """torchmd/topology.py:75-80 (non-strict >= on the + side, unlike generate_nbr_list)."""
    return -vecs.ge(0.5 * cell).to(torch.float).to(device) + vecs.lt(-0.5 * cell).to(torch.float).to(device)
