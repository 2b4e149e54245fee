"""torch.ops.mdgrad.* (csrc_torch/mdgrad_torch.cpp): the TORCH_LIBRARY op layer above the C ABI.  `get()` returns the
op namespace, or None when the library was not built or MDG_TORCH_OPS=0 -- the ctypes bindings of _lib.py then serve
the same entry points (both are the HIP kernels of libmdgrad_hip.so; there is no CPU fallback either way)."""
import os

import torch

from . import _lib

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmdgrad_torch.so")
OPS = ("nbr_build", "pair_force", "pair_hvp", "nhc_vv_forward", "nhc_vv_adjoint", "rdf_fwd", "rdf_bwd", "edge_geom",
       "edge_geom_bwd", "cfconv_fwd", "cfconv_bwd", "dense_ssp", "ssp_dual_bwd_t", "atb")
_state = {"tried": False, "ns": None}


def get():
    if not _state["tried"]:
        _state["tried"] = True
        if os.environ.get("MDG_TORCH_OPS", "1") != "0" and os.path.exists(PATH):
            _lib.load()                                   # libmdgrad_hip.so first (resolved through $ORIGIN as well)
            torch.ops.load_library(PATH)
            _state["ns"] = torch.ops.mdgrad
    return _state["ns"]


def cell_args(cs):
    """MdgCell -> the 19 numbers of the ops' `float[] cell` argument."""
    return [float(x) for x in cs.h] + [float(x) for x in cs.inv] + [float(cs.diag)]


def none_if_empty(t):
    return t if t.numel() else None
