"""GCNConv / GCN on top of ``spmm_sum`` (reference dgsparse/nn/gcnconv.py:10-70)."""
import torch
import torch.nn.functional as F

from ..spmm import spmm_sum
from ..tensor import SparseTensor
from .graph import csr_from_edge_index


class GCNConv(torch.nn.Module):
    """x' = A_hat (x W): one dense projection followed by one SpMM-sum (gcnconv.py:10-20)."""

    def __init__(self, in_size, out_size):
        super().__init__()
        self.W = torch.nn.Linear(in_size, out_size, bias=False)

    def forward(self, dcsr, x):
        return spmm_sum(dcsr, self.W(x), 0)


class GCN(torch.nn.Module):
    """Two GCNConv layers with a ReLU in between (gcnconv.py:23-35)."""

    def __init__(self, in_size, out_size, hidden_size):
        super().__init__()
        self.conv1 = GCNConv(in_size, hidden_size)
        self.conv2 = GCNConv(hidden_size, out_size)

    def forward(self, dcsr, x):
        return self.conv2(dcsr, F.relu(self.conv1(dcsr, x)))


def gcn_norm_from_edge_index(edge_index, num_nodes, add_self_loops=True):
    """Symmetric GCN normalisation D^-1/2 (A + I) D^-1/2 (gcnconv.py:36-49) -> (rowptr, col, values)."""
    rowptr, col, val = csr_from_edge_index(edge_index, num_nodes, self_loops=1.0 if add_self_loops else None)
    counts = (rowptr[1:] - rowptr[:-1]).long()
    row = torch.repeat_interleave(torch.arange(num_nodes, device=col.device), counts)
    deg = torch.zeros(num_nodes, device=col.device).index_add_(0, row, val)
    dis = deg.pow(-0.5)
    dis.masked_fill_(dis == float('inf'), 0.0)
    return rowptr, col, dis[row] * val * dis[col.long()]


def get_gcn_dcsr_from_edge_index(edge_index, num_nodes):
    """Normalised adjacency as a ``dgsparse.SparseTensor`` with trainable values (gcnconv.py:52-70)."""
    rowptr, col, val = gcn_norm_from_edge_index(edge_index, num_nodes)
    return SparseTensor(row=None, rowptr=rowptr, col=col, values=val.requires_grad_(), has_value=True)
