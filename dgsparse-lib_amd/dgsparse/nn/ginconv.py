"""GINConv / GIN on top of ``spmm_sum / spmm_max / spmm_mean`` (reference dgsparse/nn/ginconv.py:9-112)."""
import torch
import torch.nn.functional as F

from ..spmm import spmm_max, spmm_mean, spmm_sum
from ..tensor import SparseTensor
from .graph import csr_from_edge_index


class GINConv(torch.nn.Module):
    """h' = act(f((1 + eps) h + AGG_{j in N(i)} h_j)), AGG in {sum, max, mean} (ginconv.py:9-70).

    ``cached=True`` keeps the SparseTensor (CSR + CSC) built from the first ``edge_index`` - the reference accepts
    the flag but rebuilds the tensor, including a csr2csc, on every forward (ginconv.py:41-58)."""

    def __init__(self, apply_func=None, aggregator_type='sum', init_eps=0, learn_eps=False, activation=None,
                 cached=False):
        super().__init__()
        self.apply_func = apply_func
        self._aggregator_type = aggregator_type
        self.activation = activation
        self.cached = cached
        self._cached_dcsr = None
        if learn_eps:
            self.eps = torch.nn.Parameter(torch.FloatTensor([init_eps]))
        else:
            self.register_buffer('eps', torch.FloatTensor([init_eps]))

    def forward(self, edge_index, X, num_nodes):
        rst = (1 + self.eps) * X + self.aggregate_neigh(edge_index, X, num_nodes, 0)
        if self.apply_func is not None:
            rst = self.apply_func(rst)
        if self.activation is not None:
            rst = self.activation(rst)
        return rst

    def _dcsr(self, edge_index, num_nodes):
        if self.cached and self._cached_dcsr is not None:
            return self._cached_dcsr
        rowptr, col, val = csr_from_edge_index(edge_index, num_nodes)
        dcsr = SparseTensor(row=None, rowptr=rowptr, col=col, values=val.requires_grad_(), has_value=True)
        if self.cached:
            self._cached_dcsr = dcsr
        return dcsr

    def aggregate_neigh(self, edge_index, X, num_nodes, algorithm):
        dcsr = self._dcsr(edge_index, num_nodes)
        fn = {'sum': spmm_sum, 'max': spmm_max, 'mean': spmm_mean}.get(self._aggregator_type, spmm_sum)
        return fn(dcsr, X, algorithm)


class GIN(torch.nn.Module):
    """Two GINConv layers with linear update functions (ginconv.py:73-112)."""

    def __init__(self, in_size, out_size, hidden_size, aggregator_type='sum', init_eps=0, learn_eps=False,
                 activation=F.relu, cached=False):
        super().__init__()
        self.conv1 = GINConv(torch.nn.Linear(in_size, hidden_size), aggregator_type, init_eps, learn_eps, activation,
                             cached)
        self.conv2 = GINConv(torch.nn.Linear(hidden_size, out_size), aggregator_type, init_eps, learn_eps, activation,
                             cached)

    def forward(self, edge_index, X, num_nodes):
        return self.conv2(edge_index, self.conv1(edge_index, X, num_nodes), num_nodes)

    @property
    def eps(self):
        return self.conv1.eps
