"""CPU restatement of the SchNet energy the reference evaluates under GNNPotentials.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Citations: /root/reference.

The network is evaluated functionally from a state_dict with the reference's
parameter names; derivatives (forces, force-vjp) come from torch
autograd on CPU -- this is the checker, not the product.
"""
import math

import torch
import torch.nn.functional as Fnn

from .md_oracle import nbr_list, cell_matrix

__all__ = ["schnet_param_names", "schnet_energy", "SchNetTerm", "ssp"]

LOG2 = math.log(2.0)


def ssp(x):
    """shifted_softplus (nff/nn/activations.py:5-11)."""
    return Fnn.softplus(x) - LOG2


def schnet_param_names(sd):
    """Trainable tensors in nn.Module.parameters() order (GaussianSmearing width/offsets are
    buffers when trainable_gauss=False: nff/nn/layers.py:62-67)."""
    return [k for k in sd if not (k.endswith(".0.width") or k.endswith(".0.offsets"))]


def schnet_energy(sd, z, xyz, nbr, offsets):
    """SchNet.forward -> summed 'energy' (nff/nn/models/schnet.py:113-171).

    distances use the RAW image flags, not offsets@cell (schnet.py:140-142 with
    torchmd/interface.py:122-123 -- the quirk of SURVEY 0.8)."""
    a0, a1 = nbr[:, 0], nbr[:, 1]
    e = (xyz[a0] - xyz[a1] - offsets.to(xyz)).pow(2).sum(1).sqrt()[:, None]      # :142
    r = sd["atom_embed.weight"][z]                                                # :146
    n_conv = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("convolutions."))
    for l in range(n_conv):
        pre = "convolutions.%d.moduledict." % l
        mu = sd[pre + "message_edge_filter.0.offsets"]
        wd = sd[pre + "message_edge_filter.0.width"]
        g = torch.exp(-0.5 / wd.pow(2) * (e - mu).pow(2))                         # layers.py:19-29
        W = Fnn.linear(ssp(Fnn.linear(g, sd[pre + "message_edge_filter.1.weight"],
                                      sd[pre + "message_edge_filter.1.bias"])),
                       sd[pre + "message_edge_filter.3.weight"],
                       sd[pre + "message_edge_filter.3.bias"])                     # modules.py:531-541
        h = Fnn.linear(r, sd[pre + "message_node_filter.weight"],
                       sd[pre + "message_node_filter.bias"])                      # modules.py:542,564
        m = torch.zeros(r.shape[0], W.shape[1], dtype=r.dtype)
        m = m.index_add(0, a1, h[a0] * W)                                         # graphconv.py:49
        m = m.index_add(0, a0, h[a1] * W)                                         # graphconv.py:51
        dr = Fnn.linear(ssp(Fnn.linear(m, sd[pre + "update_function.0.weight"],
                                       sd[pre + "update_function.0.bias"])),
                        sd[pre + "update_function.2.weight"],
                        sd[pre + "update_function.2.bias"])                       # modules.py:543-547
        r = r + dr                                                                # schnet.py:149-151
    ro = "atomwisereadout.readout.energy."
    out = Fnn.linear(ssp(Fnn.linear(r, sd[ro + "linear0.weight"], sd[ro + "linear0.bias"])),
                     sd[ro + "linear2.weight"], sd[ro + "linear2.bias"])          # nn/utils.py:56-75
    return out.sum()                                                              # graphop.py:25-30


class SchNetTerm:
    """GNNPotentials (torchmd/interface.py:86-136) as an oracle term."""

    def __init__(self, sd, z, cutoff, cell, ex_pairs=None):
        self.sd = {k: torch.as_tensor(v) for k, v in sd.items()}
        self.names = schnet_param_names(self.sd)
        self.z = torch.as_tensor(z, dtype=torch.long)
        self.cutoff, self.cell, self.ex_pairs = float(cutoff), cell_matrix(cell), ex_pairs
        self.nbr = self.off = None

    @property
    def n_theta(self):
        return sum(self.sd[k].numel() for k in self.names)

    def reset(self, q):                                        # interface.py:116-123
        self.nbr, self.off = nbr_list(q.detach(), self.cutoff, self.cell.to(q), None, self.ex_pairs)

    def _sd(self, dtype, grad):
        out = {}
        for k, v in self.sd.items():
            v = v.to(dtype) if v.is_floating_point() else v
            out[k] = v.detach().clone().requires_grad_(True) if (grad and k in self.names) else v
        return out

    def energy(self, q):
        return schnet_energy(self._sd(q.dtype, False), self.z, q, self.nbr, self.off)

    def force(self, q):
        with torch.enable_grad():
            x = q.detach().requires_grad_(True)
            U = schnet_energy(self._sd(q.dtype, False), self.z, x, self.nbr, self.off)
            (g,) = torch.autograd.grad(U, x)
        return -g

    def force_vjp(self, q, w):
        with torch.enable_grad():
            sd = self._sd(q.dtype, True)
            x = q.detach().requires_grad_(True)
            U = schnet_energy(sd, self.z, x, self.nbr, self.off)
            (g,) = torch.autograd.grad(U, x, create_graph=True)
            ps = [sd[k] for k in self.names]
            grads = torch.autograd.grad((w.detach() * (-g)).sum(), [x] + ps, allow_unused=True)
        dth = torch.cat([(gr if gr is not None else torch.zeros_like(p)).reshape(-1)
                         for gr, p in zip(grads[1:], ps)])
        return (-g).detach(), grads[0].detach(), dth.detach()
