"""GCN layers on ``spmm_sum`` and the symmetric GCN normalisation, torch_sparse-free.  Names and signatures follow the
reference's ``dgsparse/nn/gcnconv.py`` (GCNConv :10-20, GCN :23-35, gcn_norm_from_edge_index :36-49,
get_gcn_dcsr_from_edge_index :52-70)."""
import torch
import torch.nn.functional as F

from ..spmm import spmm_sum
from ..tensor import SparseTensor
from .fused import spmm_sum_fused
from .graph import csr_from_edge_index


class GCNConv(torch.nn.Module):
    """``x' = A_hat @ (x W)``: dense projection first, then one SpMM-sum with the normalised adjacency.
    ``activation='relu'`` (not in the reference's layer) fuses the ReLU that usually follows into the SpMM's row-end store:
    same values bit for bit, one pass over the output less."""

    def __init__(self, in_size, out_size, activation=None):
        super().__init__()
        if activation not in (None, 'relu'):
            raise ValueError(activation)
        self.W = torch.nn.Linear(in_size, out_size, bias=False)
        self.activation = activation

    def forward(self, dcsr, x):
        projected = self.W(x)
        if self.activation == 'relu' and projected.is_cuda:
            return spmm_sum_fused(dcsr, projected, relu=True)
        out = spmm_sum(dcsr, projected, 0)
        return F.relu(out) if self.activation == 'relu' else out


class GCN(torch.nn.Module):
    """conv -> ReLU -> conv."""

    def __init__(self, in_size, out_size, hidden_size):
        super().__init__()
        self.conv1 = GCNConv(in_size, hidden_size, activation='relu')  # the ReLU rides in the SpMM's epilogue
        self.conv2 = GCNConv(hidden_size, out_size)

    def forward(self, dcsr, x):
        hidden = self.conv1(dcsr, x)
        return self.conv2(dcsr, hidden)


def gcn_norm_from_edge_index(edge_index, num_nodes, add_self_loops=True):
    """``D^-1/2 (A + I) D^-1/2`` as CSR ``(rowptr, col, values)``; D = weighted row degree of ``A + I``; nodes of
    degree 0 get a zero scale (the reference masks the inf of ``deg ** -0.5`` the same way)."""
    rowptr, col, w = csr_from_edge_index(edge_index, num_nodes, self_loops=1.0 if add_self_loops else None)
    src = torch.repeat_interleave(torch.arange(num_nodes, device=col.device), (rowptr[1:] - rowptr[:-1]).long())
    degree = torch.zeros(num_nodes, device=col.device).index_add_(0, src, w)
    scale = torch.where(degree > 0, degree.rsqrt(), torch.zeros_like(degree))
    return rowptr, col, scale[src] * w * scale[col.long()]


def get_gcn_dcsr_from_edge_index(edge_index, num_nodes):
    """The normalised adjacency as a ``dgsparse.SparseTensor`` whose values require grad."""
    rowptr, col, w = gcn_norm_from_edge_index(edge_index, num_nodes)
    return SparseTensor(rowptr=rowptr, col=col, values=w.requires_grad_(), has_value=True)
