"""GIN layers on the SpMM operators.  Same constructor arguments, attribute names and call signatures as the
reference's ``dgsparse/nn/ginconv.py`` (GINConv :9-70, GIN :73-112); torch_sparse is not needed."""
import torch
import torch.nn.functional as F

from ..spmm import spmm_max, spmm_mean, spmm_sum
from ..tensor import SparseTensor
from .graph import csr_from_edge_index

_AGGREGATORS = {'sum': spmm_sum, 'max': spmm_max, 'mean': spmm_mean}


class GINConv(torch.nn.Module):
    r"""``h_i' = act( f( (1 + eps) h_i + AGG_{j in N(i)} h_j ) )`` with AGG in {sum, max, mean}.

    ``cached=True`` builds the neighbourhood ``SparseTensor`` (CSR + its CSC view + its locality plans) once and reuses
    it for as long as the SAME ``edge_index`` tensor (object, version) and ``num_nodes`` come in; the reference accepts
    the flag but converts ``edge_index`` -- including a csr2csc -- on every forward."""

    def __init__(self, apply_func=None, aggregator_type='sum', init_eps=0, learn_eps=False, activation=None,
                 cached=False):
        super().__init__()
        self.apply_func, self.activation = apply_func, activation
        self._aggregator_type = aggregator_type
        self.cached, self._cached_dcsr, self._cached_key = cached, None, None
        eps = torch.FloatTensor([init_eps])
        if learn_eps:
            self.eps = torch.nn.Parameter(eps)
        else:
            self.register_buffer('eps', eps)

    def _neighbourhood(self, edge_index, num_nodes) -> SparseTensor:
        key = (id(edge_index), edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), int(num_nodes))
        if self._cached_dcsr is not None and self._cached_key == key:
            return self._cached_dcsr
        rowptr, col, w = csr_from_edge_index(edge_index, num_nodes)
        dcsr = SparseTensor(rowptr=rowptr, col=col, values=w.requires_grad_(), has_value=True)
        if self.cached:
            self._cached_dcsr, self._cached_key = dcsr, key
        return dcsr

    def aggregate_neigh(self, edge_index, X, num_nodes, algorithm):
        agg = _AGGREGATORS.get(self._aggregator_type, spmm_sum)  # unknown names fall back to sum, as the reference
        return agg(self._neighbourhood(edge_index, num_nodes), X, algorithm)

    def forward(self, edge_index, X, num_nodes):
        out = (1 + self.eps) * X + self.aggregate_neigh(edge_index, X, num_nodes, 0)
        for stage in (self.apply_func, self.activation):
            if stage is not None:
                out = stage(out)
        return out


class GIN(torch.nn.Module):
    """Two GINConv layers whose update functions are single Linear layers."""

    def __init__(self, in_size, out_size, hidden_size, aggregator_type='sum', init_eps=0, learn_eps=False,
                 activation=F.relu, cached=False):
        super().__init__()
        shared = (aggregator_type, init_eps, learn_eps, activation, cached)
        self.conv1 = GINConv(torch.nn.Linear(in_size, hidden_size), *shared)
        self.conv2 = GINConv(torch.nn.Linear(hidden_size, out_size), *shared)

    def forward(self, edge_index, X, num_nodes):
        hidden = self.conv1(edge_index, X, num_nodes)
        return self.conv2(edge_index, hidden, num_nodes)

    @property
    def eps(self):
        return self.conv1.eps
