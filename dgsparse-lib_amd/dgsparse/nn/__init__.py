"""GNN layers that call the SpMM operators (reference dgsparse/nn/gcnconv.py, ginconv.py), torch_sparse-free."""
from .gcnconv import GCN, GCNConv, gcn_norm_from_edge_index, get_gcn_dcsr_from_edge_index
from .fused import spmm_sum_fused
from .ginconv import GIN, GINConv
from .graph import csr_from_edge_index

__all__ = ['GCNConv', 'GCN', 'GINConv', 'GIN', 'gcn_norm_from_edge_index', 'get_gcn_dcsr_from_edge_index',
           'csr_from_edge_index', 'spmm_sum_fused']
