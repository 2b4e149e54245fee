"""`MessagePassingModule` (nff/nn/graphconv.py:11-53): message -> aggregate over both directions of the undirected list ->
update, under the reference's method names so that user subclasses written against it keep working.  `SchNetConv` derives from
it; with a topology attached it replaces message + aggregate by one HIP gather (nn/schnet.py)."""
import torch.nn as nn

from .graphop import scatter_add


class MessagePassingModule(nn.Module):
    def __init__(self):
        super().__init__()

    def message(self, r, e, a, aggr_wgt):
        """(m_ij, m_ji): the node rows at either end of every pair times the pair's edge row."""
        assert r.shape[-1] == e.shape[-1]
        if aggr_wgt is not None:
            r = r * aggr_wgt
        return r[a[:, 0]] * e, r[a[:, 1]] * e

    def aggregate(self, message, index, size):
        return scatter_add(src=message, index=index, dim=0, dim_size=size)

    def update(self, r):
        return r

    def forward(self, r, e, a, aggr_wgt=None):
        n = r.shape[0]
        m_ij, m_ji = self.message(r, e, a, aggr_wgt)
        out = self.aggregate(m_ij, a[:, 1], n)            # i -> j
        out = out + self.aggregate(m_ji, a[:, 0], n)      # j -> i
        return self.update(out)
