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
    F = frames.shape[0]
    # all frames in ONE list build: the frames are groups of one stacked system (pairs never cross groups), so the
    # half list comes out sorted by (frame, i, j) -- the reference's order -- with one host sync for the pair count
    ell = ops.build_ell(frames.reshape(F * N, 3), cs, cutoff, mask, group=N if F > 1 else None)
    nbr, off = ell.half_list()
    if get_dis:
        flat = frames.reshape(F * N, 3)
        dis = compute_dis(flat, nbr, off, cellm).reshape(-1)
    if xyz.dim() > 2:
        frame = torch.div(nbr[:, :1], N, rounding_mode="floor")
        nbr = torch.cat([frame, nbr - frame * N], 1)
    if get_dis:
        return nbr, dis, off
    return nbr, off


def get_offsets(vecs, cell, device):
    """torchmd/topology.py:75-80 (non-strict >= on the + side, unlike generate_nbr_list)."""
    return -vecs.ge(0.5 * cell).to(torch.float).to(device) + vecs.lt(-0.5 * cell).to(torch.float).to(device)
