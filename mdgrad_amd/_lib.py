"""ctypes binding of libmdgrad_hip.so (the C ABI declared in include/mdgrad_hip.h).

There is NO fallback: if the HIP library is missing this module raises, and every hot-path
op in the package goes through it.  PyTorch is used only for device memory, streams and
autograd bookkeeping; pointers handed to the library are `tensor.data_ptr()`.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MDG_LIB: another build of the same library, e.g. a compile-time variant of one kernel file for an A/B run -- tools/variants.sh)
LIB_PATH = os.environ.get("MDG_LIB") or os.path.join(_HERE, "lib", "libmdgrad_hip.so")

MAX_TERMS, MAX_THETA, MAX_CHAINS = 4, 3, 16
PAIR_LJ, PAIR_MORSE, PAIR_BUCK, PAIR_YUKAWA = 0, 1, 2, 3


class MdgPairTerm(C.Structure):
    _fields_ = [("kind", C.c_int32), ("p", C.c_int32), ("q", C.c_int32), ("c", C.c_float),
                ("a", C.c_float), ("phi", C.c_float), ("cutoff", C.c_float),
                ("theta_off", C.c_int32), ("n_theta", C.c_int32), ("reserved", C.c_int32),
                ("mask", C.c_void_p)]


class MdgTerms(C.Structure):
    _fields_ = [("n_terms", C.c_int32), ("n_theta_total", C.c_int32), ("t", MdgPairTerm * MAX_TERMS)]


class MdgCell(C.Structure):
    _fields_ = [("h", C.c_float * 9), ("inv", C.c_float * 9), ("diag", C.c_int32)]


class MdgTrajParams(C.Structure):
    _fields_ = [("n_rep", C.c_int32), ("n_atoms", C.c_int32), ("n_frames", C.c_int32),
                ("n_chains", C.c_int32), ("ensemble", C.c_int32), ("block", C.c_int32),
                ("T", C.c_float), ("n_dof", C.c_float), ("Q", C.c_float * MAX_CHAINS)]


class MdgRdfFuse(C.Structure):
    """Fused RDF observable of the wave-per-replica trajectory kernels (include/mdgrad_hip.h)."""
    _fields_ = [("mu", C.c_void_p), ("nbins", C.c_int32), ("coeff", C.c_float), ("mu0", C.c_float), ("spacing", C.c_float),
                ("cutoff", C.c_float), ("frame_start", C.c_int32), ("frame_stride", C.c_int32)]


class MdgFilterNet(C.Structure):
    """Host struct of device pointers describing one SchNet filter network (include/mdgrad_hip.h)."""
    _fields_ = [("mu", C.c_void_p), ("coef", C.c_void_p), ("W1", C.c_void_p), ("b1", C.c_void_p),
                ("W2", C.c_void_p), ("b2", C.c_void_p), ("n_gauss", C.c_int32), ("n_filters", C.c_int32)]


class MdgGradJob(C.Structure):
    """One reduction of mdg_grad_jobs (include/mdgrad_hip.h)."""
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("A2", C.c_void_p), ("B2", C.c_void_p), ("row_map", C.c_void_p),
                ("rows", C.c_int64), ("m", C.c_int32), ("n", C.c_int32), ("kind", C.c_int32), ("pad_", C.c_int32),
                ("out_off", C.c_int64)]


GRAD_ATB, GRAD_COLSUM, GRAD_AXPY, GRAD_JOBS_MAX = 0, 1, 2, 32


class MdgChainStage(C.Structure):
    """One Dense stage of mdg_row_chain (include/mdgrad_hip.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("W", "bias", "in0", "in1", "res0", "res1", "aux0", "aux1", "out0", "out1", "sig",
                                          "pre0", "pre1", "out0_h", "out1_h")] + \
               [(n, C.c_int32) for n in ("K", "M", "trans", "act", "mode", "pad_")]


SCHNET_MAX_LAYERS = 8


class MdgSchnetLayer(C.Structure):
    """One interaction block of MdgSchnetPlan (include/mdgrad_hip.h)."""
    _fields_ = [("filt", MdgFilterNet)] + [(n, C.c_void_p) for n in ("Wn", "bn", "U1", "c1", "U2", "c2")] + \
               [("off_" + n, C.c_int64) for n in ("W1", "b1", "W2", "b2", "Wn", "bn", "U1", "c1", "U2", "c2")] + \
               [(n, C.c_int32) for n in ("bf16", "bf16_rev", "rows16", "b2col")]


class MdgSchnetPlan(C.Structure):
    """A SchNet network + topology + workspace for mdg_schnet_force / mdg_schnet_force_vjp (include/mdgrad_hip.h)."""
    _fields_ = [(n, C.c_int32) for n in ("n_atoms", "n_layers", "n_atom_basis", "n_readout")] + \
               [("layer", MdgSchnetLayer * SCHNET_MAX_LAYERS)] + [(n, C.c_void_p) for n in ("L1", "l1", "L2")] + \
               [(n, C.c_int64) for n in ("off_L1", "off_l1", "off_L2", "off_embed")] + \
               [(n, C.c_void_p) for n in ("r0", "h0", "h0_16", "onehot", "uniq")] + [("n_species", C.c_int32), ("masked", C.c_int32)] + \
               [("nbr", C.c_void_p), ("offsets", C.c_void_p), ("n_edges", C.c_int64)] + \
               [(n, C.c_void_p) for n in ("col", "eid", "cnt", "n_valid")] + [("max_nbr", C.c_int32), ("cutoff", C.c_float)] + \
               [("cell", MdgCell), ("ws", C.c_void_p), ("ws_floats", C.c_int64), ("stash", C.c_int32), ("chain_x3", C.c_int32)]


CHAIN_NONE, CHAIN_MUL, CHAIN_HEAD, CHAIN_SSP_BWD, CHAIN_MAX_STAGES, CHAIN_MAX_WIDTH = 0, 1, 2, 3, 8, 512
BONDED_BOND, BONDED_ANGLE = 0, 1                 # include/mdgrad_hip.h MDG_BONDED_*
CFCONV_BF16, CFCONV_ROWS16 = 1, 2                # include/mdgrad_hip.h MDG_CFCONV_*

P = C.c_void_p
_SIGNATURES = {
    "mdg_last_error": (C.c_char_p, []),
    "mdg_version": (C.c_int, []),
    "mdg_nbr_build_dense": (C.c_int, [P, C.c_int, C.POINTER(MdgCell), C.c_float, P, P, P, P, C.c_int, P, P]),
    "mdg_nbr_build_dense_groups": (C.c_int, [P, C.c_int, C.c_int, C.POINTER(MdgCell), C.c_float, P, P, P, P,
                                             C.c_int, P, P]),
    "mdg_nbr_cell_scratch": (C.c_int64, [C.c_int, C.POINTER(MdgCell), C.c_float]),
    "mdg_nbr_build_cell": (C.c_int, [P, C.c_int, C.POINTER(MdgCell), C.c_float, P, P, P, P, C.c_int, P, P, P]),
    "mdg_nbr_cell_scratch_groups": (C.c_int64, [C.c_int, C.c_int, C.POINTER(MdgCell), C.c_float]),
    "mdg_nbr_build_cell_groups": (C.c_int, [P, C.c_int, C.c_int, C.POINTER(MdgCell), C.c_float, P, P, P, P, C.c_int,
                                            P, P, P]),
    "mdg_nbr_half_count": (C.c_int, [P, P, C.c_int, C.c_int, P, P]),
    "mdg_nbr_half_fill": (C.c_int, [P, P, P, P, C.c_int, C.c_int, P, P, P, P]),
    "mdg_nbr_half_fill_padded": (C.c_int, [P, P, P, P, C.c_int, C.c_int, C.c_int64, C.c_float, P, P, P, P, P, P]),
    "mdg_nbr_verlet_rebuild": (C.c_int, [P, C.c_int, C.c_int, C.POINTER(MdgCell), C.c_float, C.c_float, P, C.c_int, P, P, P,
                                         C.c_int, C.c_int64, C.c_float, P, P, P, P, P, P, P, P, P, P]),
    "mdg_edge_geom_prepare": (C.c_int, [P, P, P, P, C.c_int64, C.POINTER(MdgCell), C.c_float, P, P, P, P, P, C.c_int64, P]),
    "mdg_edge_geom_masked": (C.c_int, [P, P, P, P, C.c_int64, C.POINTER(MdgCell), C.c_float, P, P, P, P, P]),
    "mdg_pair_partial_size": (C.c_int64, [C.c_int]),
    "mdg_pair_eval_ell": (C.c_int, [P, C.c_int, C.POINTER(MdgCell), P, P, P, C.c_int,
                                    C.POINTER(MdgPairTerm), P, P, P, P, P, P, P, P, P]),
    "mdg_pair_eval_ell_into": (C.c_int, [P, C.c_int, C.POINTER(MdgCell), P, P, P, C.c_int,
                                         C.POINTER(MdgPairTerm), P, P, P, P, P, P, P, P, C.c_float, C.c_int, P]),
    "mdg_traj_fwd_small": (C.c_int, [C.POINTER(MdgTrajParams), C.POINTER(MdgCell), C.POINTER(MdgTerms),
                                     P, P, P, P, P, P, P, P, P, P, P]),
    "mdg_traj_adj_small": (C.c_int, [C.POINTER(MdgTrajParams), C.POINTER(MdgCell), C.POINTER(MdgTerms),
                                     P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "mdg_traj_rdf_supported": (C.c_int, [C.POINTER(MdgTrajParams), C.POINTER(MdgCell), C.POINTER(MdgTerms),
                                         C.POINTER(MdgRdfFuse)]),
    "mdg_traj_fwd_small_rdf": (C.c_int, [C.POINTER(MdgTrajParams), C.POINTER(MdgCell), C.POINTER(MdgTerms),
                                         P, P, P, P, P, P, P, P, P, P, C.POINTER(MdgRdfFuse), P, P]),
    "mdg_traj_adj_small_rdf": (C.c_int, [C.POINTER(MdgTrajParams), C.POINTER(MdgCell), C.POINTER(MdgTerms),
                                         P, P, P, P, P, P, P, P, P, P, P, P, P, C.POINTER(MdgRdfFuse), P, P]),
    "mdg_traj_large_workspace": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "mdg_traj_large_list_builds": (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, P, P]),
    "mdg_traj_fwd_large": (C.c_int, [C.POINTER(MdgTrajParams), C.POINTER(MdgCell), C.POINTER(MdgTerms),
                                     P, P, P, P, P, P, P, P, P, P, P, P]),
    "mdg_traj_adj_large": (C.c_int, [C.POINTER(MdgTrajParams), C.POINTER(MdgCell), C.POINTER(MdgTerms),
                                     P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "mdg_traj_large_stale_words": (C.c_int64, [C.c_int, C.c_int]),
    "mdg_traj_fwd_large_stale": (C.c_int, [C.POINTER(MdgTrajParams), C.POINTER(MdgCell), C.POINTER(MdgTerms),
                                           P, P, P, P, P, P, P, P, P, P, P, C.c_int, C.c_int64, P, P]),
    "mdg_traj_adj_large_stale": (C.c_int, [C.POINTER(MdgTrajParams), C.POINTER(MdgCell), C.POINTER(MdgTerms),
                                           P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, C.c_int, C.c_int64, P, P]),
    "mdg_rdf_partial_size": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "mdg_rdf_fwd": (C.c_int, [P, C.c_int, C.c_int, C.POINTER(MdgCell), C.c_float, P, P, C.c_float,
                              C.c_int, P, P, P]),
    "mdg_rdf_fwd_uniform": (C.c_int, [P, C.c_int, C.c_int, C.POINTER(MdgCell), C.c_float, P, P, C.c_float, C.c_float,
                                      C.c_int, P, P, P]),
    "mdg_rdf_ell_supported": (C.c_int, [C.c_float, C.c_float, C.c_int]),
    "mdg_rdf_fwd_ell": (C.c_int, [P, C.c_int64, C.POINTER(MdgCell), P, P, P, C.c_int, P, C.c_float, C.c_float, C.c_int, P, P]),
    "mdg_rdf_cell_supported": (C.c_int, [C.c_int, C.POINTER(MdgCell), C.c_float]),
    "mdg_rdf_cell_scratch": (C.c_int64, [C.c_int, C.c_int, C.POINTER(MdgCell), C.c_float]),
    "mdg_rdf_fwd_cell": (C.c_int, [P, C.c_int, C.c_int, C.POINTER(MdgCell), C.c_float, P, C.c_float, C.c_float, C.c_int,
                                   P, P, P]),
    "mdg_rdf_bwd_cell": (C.c_int, [C.c_int, C.c_int, C.POINTER(MdgCell), C.c_float, C.POINTER(MdgPairTerm), P, P, P, P]),
    "mdg_rdf_bwd": (C.c_int, [P, C.c_int, C.c_int, C.POINTER(MdgCell), C.c_float, P, P, C.c_float,
                              C.c_int, P, P, P]),
    "mdg_rdf_bwd_uniform": (C.c_int, [P, C.c_int, C.c_int, C.POINTER(MdgCell), C.c_float, P, P, C.c_float, C.c_float,
                                      C.c_int, P, P, P]),
    "mdg_nhc_rhs": (C.c_int, [P, P, P, P, P, P, C.c_float, C.c_int, C.c_int, C.c_int, P, P, P]),
    "mdg_nhc_vjp": (C.c_int, [P, P, P, P, P, P, P, C.c_int, C.c_int, C.c_int, P, P, P]),
    "mdg_row_chain": (C.c_int, [P, C.c_int, C.c_int, C.c_int, P]),      # (MdgChainStage*: an array object or a raw address)
    "mdg_nhv_scratch_floats": (C.c_int64, [C.c_int, C.c_int]),
    "mdg_nhv_kick": (C.c_int, [P, P, P, P, P, P, P, C.c_float, P, P, C.c_int, C.c_int, C.c_int, P, P, P, P, P]),
    "mdg_nhv_finish": (C.c_int, [P, P, P, P, P, P, P, P, P, P, P, C.c_float, P, P, C.c_int, C.c_int, C.c_int, C.c_int, P, P, P, P,
                                 P]),
    "mdg_nhv_adj_pre": (C.c_int, [P, P, P, P, P, P, C.c_int, C.c_int, C.c_int, P, P, P, P, P]),
    "mdg_nhv_adj_mid": (C.c_int, [P, P, P, P, P, P, P, P, P, P, P, C.c_float, P, P, C.c_int, C.c_int, C.c_int,
                                  P, P, P, P, P, P, P, P, P]),
    "mdg_nhv_adj_end": (C.c_int, [P, P, P, P, P, P, P, P, P, P, C.c_int, P, P, P, C.c_int, C.c_int, C.c_int, P, P, P, P, P]),
    "mdg_edge_diff": (C.c_int, [P, P, C.c_int64, C.c_int, P, P]),
    "mdg_edge_scatter": (C.c_int, [P, P, P, P, C.c_int, C.c_int, C.c_int, P, P]),
    "mdg_cfconv_agg": (C.c_int, [P, P, P, P, P, C.c_int, C.c_int, C.c_int, P, P]),
    "mdg_edge_prod": (C.c_int, [P, P, P, C.c_int64, C.c_int, P, P]),
    "mdg_smear": (C.c_int, [P, P, P, C.c_int64, C.c_int, P, P, P]),
    "mdg_ssp": (C.c_int, [P, C.c_int64, P, P, P]),
    "mdg_mul_row": (C.c_int, [P, P, P, C.c_int64, C.c_int, P, P]),
    "mdg_ssp_dual_bwd": (C.c_int, [P, P, P, P, C.c_int64, P, P, P]),
    "mdg_ssp_dual_bwd_t": (C.c_int, [P, P, P, P, C.c_int64, P, P, P]),
    "mdg_readout_head": (C.c_int, [P, P, P, C.c_int64, C.c_int, P, P, P]),
    "mdg_smear_bwd": (C.c_int, [P, P, P, P, P, P, C.c_int64, C.c_int, P, P, P]),
    "mdg_atb_workspace": (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
    "mdg_atb": (C.c_int, [P, P, C.c_int64, C.c_int, C.c_int, P, P, P]),
    "mdg_atb2": (C.c_int, [P, P, P, P, C.c_int64, C.c_int, C.c_int, P, P, P]),
    "mdg_grad_jobs_workspace": (C.c_int64, [P, C.c_int]),
    "mdg_grad_jobs": (C.c_int, [P, C.c_int, P, C.c_float, P, P, C.c_int, P, P]),
    "mdg_vacf_workspace": (C.c_int64, [C.c_int]),
    "mdg_vacf_fwd": (C.c_int, [P, C.c_int, C.c_int64, C.c_int, P, P, P]),
    "mdg_vacf_bwd": (C.c_int, [P, P, C.c_int, C.c_int64, C.c_int, P, P]),
    "mdg_temperature": (C.c_int, [P, P, C.c_int, C.c_int, C.c_float, P, P]),
    "mdg_cfconv_supported": (C.c_int, [C.c_int, C.c_int]),
    "mdg_edge_geom": (C.c_int, [P, P, P, P, C.c_int64, P, P, P, P, P]),
    "mdg_edge_geom_bwd": (C.c_int, [P, P, P, P, P, P, P, P, P, C.c_int, C.c_int, P, P, P]),
    "mdg_cfconv_fwd": (C.c_int, [C.POINTER(MdgFilterNet), P, P, P, P, P, P, P, C.c_int, C.c_int, P, P, P, P, P]),
    "mdg_cfconv_fwd_bf16": (C.c_int, [C.POINTER(MdgFilterNet), P, P, P, P, P, P, P, C.c_int, C.c_int, P, P, P, P, P]),
    "mdg_cfconv_bwd_workspace": (C.c_int64, [C.c_int, C.c_int, C.c_int64]),
    "mdg_cfconv_bwd": (C.c_int, [C.POINTER(MdgFilterNet), P, P, P, C.c_int64, P, P, P, P, P, P, P, P, P, P, P, P]),
    "mdg_cfconv_bwd_bf16": (C.c_int, [C.POINTER(MdgFilterNet), P, P, P, C.c_int64, P, P, P, P, P, P, P, P, P, P, P, P]),
    "mdg_cfconv_bwd_smear": (C.c_int, [C.POINTER(MdgFilterNet), P, P, P, C.c_int64, P, P, P, P, P, P, P, P, P, P, P, P, P,
                                       C.c_int, P]),
    "mdg_cfconv_rows16_supported": (C.c_int, [C.c_int, C.c_int]),
    "mdg_cfconv_fwd_rows16": (C.c_int, [C.POINTER(MdgFilterNet), P, P, P, P, P, P, P, C.c_int, C.c_int, P, P, P, P, P]),
    "mdg_cfconv_bwd_rows16": (C.c_int, [C.POINTER(MdgFilterNet), P, P, P, C.c_int64, C.c_int, P, P, P, P, P, P, P, P, P, P, P, P,
                                        P, P]),
    "mdg_rows_to_bf16": (C.c_int, [P, C.c_int64, C.c_int, C.c_int, P, P]),
    "mdg_cfconv_stash_width": (C.c_int, [C.c_int]),
    "mdg_cfconv_filter_stash": (C.c_int, [C.POINTER(MdgFilterNet), P, P, C.c_int64, P, P, P, P]),
    "mdg_cfconv_fwd_stashed": (C.c_int, [C.POINTER(MdgFilterNet), P, P, P, P, P, P, P, P, C.c_int, C.c_int, P, P, C.c_int, P]),
    "mdg_cfconv_bias_column": (C.c_int, [C.c_int]),
    "mdg_cfconv_bwd_theta": (C.c_int, [C.POINTER(MdgFilterNet), P, P, P, C.c_int64, C.c_int, P, P, P, P, P, P, P, P, P, P, P, P,
                                       P, P, C.c_int, P]),
    "mdg_dense": (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P, P, P, P, P, P, P, P, P, P]),
    "mdg_cfconv_filter": (C.c_int, [P, C.c_int64, P, P, C.c_int, P, P, P, P, C.c_int, P, P]),
    "mdg_cfconv_filter_bf16": (C.c_int, [P, C.c_int64, P, P, C.c_int, P, P, P, P, C.c_int, P, P]),
    "mdg_traj_stale_words": (C.c_int64, [C.c_int, C.c_int]),
    "mdg_traj_fwd_small_stale": (C.c_int, [C.POINTER(MdgTrajParams), C.POINTER(MdgCell), C.POINTER(MdgTerms), P, P, P, P, P, P,
                                           P, P, P, P, C.c_int, C.c_int64, P, P]),
    "mdg_traj_adj_small_stale": (C.c_int, [C.POINTER(MdgTrajParams), C.POINTER(MdgCell), C.POINTER(MdgTerms), P, P, P, P, P, P,
                                           P, P, P, P, P, P, P, C.c_int, C.c_int64, P, P]),
    "mdg_schnet_plan_sizeof": (C.c_int64, []),
    "mdg_schnet_workspace": (C.c_int64, [C.POINTER(MdgSchnetPlan), C.c_int, C.c_int]),
    "mdg_schnet_force": (C.c_int, [C.POINTER(MdgSchnetPlan), P, P, P, P]),
    "mdg_schnet_force_vjp": (C.c_int, [C.POINTER(MdgSchnetPlan), P, P, P, P, P, C.c_float, P, P, P, P]),
    "mdg_bonded_eval": (C.c_int, [P, C.c_int, C.POINTER(C.c_float), C.c_int, P, C.c_int, C.c_float, C.c_float, P, P, P, P, P, P,
                                  C.c_float, C.c_int, P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)
_lib = None


class MdgradLibraryError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises MdgradLibraryError when the shared
    library has not been built -- there is deliberately no CPU / pure-torch fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MdgradLibraryError(
            "libmdgrad_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950).  mdgrad_amd has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)          # torch is imported above: its libamdhip64 is already resident
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().mdg_last_error()
        raise RuntimeError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def ptr(t):
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError("mdgrad_amd: non-contiguous tensor passed to the HIP library")
    return C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    """The current HIP stream of `device` (torch.device / index / None) as a void*: one call into torch's C layer
    (torch.cuda.current_stream() builds a Stream object per call -- ~5 us, several hundred times per MD pass)."""
    if device is None:
        idx = torch.cuda.current_device()
    elif isinstance(device, int):
        idx = device
    else:
        idx = device.index if isinstance(device, torch.device) else torch.device(device).index
        if idx is None:
            idx = torch.cuda.current_device()
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(idx))


def require_gpu(t, name="tensor", dtype=torch.float32):
    if not t.is_cuda:
        raise RuntimeError("mdgrad_amd: %s must live on a HIP device (got %s); the hot path has no CPU "
                           "implementation" % (name, t.device))
    if dtype is not None and t.dtype != dtype:
        raise TypeError("mdgrad_amd: %s must be %s (got %s)" % (name, dtype, t.dtype))
    return t


def make_cell(cell):
    """MdgCell from a [3] or [3,3] tensor/array (host side).  The inverse is computed with
    torch (fp32, CPU) like the reference's `cell.inverse()` (torchmd/topology.py:59)."""
    c = torch.as_tensor(cell, dtype=torch.float32).detach().cpu()
    if c.dim() == 1:
        c = torch.diag(c)
    inv = c.inverse()
    mc = MdgCell()
    flat, finv = c.reshape(-1).tolist(), inv.reshape(-1).tolist()
    for k in range(9):
        mc.h[k] = flat[k]
        mc.inv[k] = finv[k]
    off = [flat[k] for k in (1, 2, 3, 5, 6, 7)]
    mc.diag = int(all(x == 0.0 for x in off))
    return mc
