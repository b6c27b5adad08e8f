"""``spmm_sum / spmm_mean / spmm_max / spmm_min`` -- the reference's public operators
(dgsparse/spmm.py:5,31,57,83): same names, same ``(sparse, dense, algorithm)`` arguments, same unpacking of
the SparseTensor into the nine-argument ``torch.ops.dgsparse_spmm.*`` call."""
import torch

from . import _capi
from .tensor import SparseTensor


def _call(op, code, sparse: SparseTensor, dense: torch.Tensor, algorithm) -> torch.Tensor:
    st = sparse.storage
    if dense.dim() != 2 or dense.shape[0] < st.sparse_sizes[1]:
        raise ValueError(f'dgsparse: dense has shape {tuple(dense.shape)} but the sparse tensor references '
                         f'{st.sparse_sizes[1]} columns')
    values = st.values()
    cuda = dense.is_cuda and st.col().is_cuda
    plan, pinfo = st.spmm_plan('csr', dense.shape[1]) if cuda else (None, None)
    if cuda and code in (_capi.SUM, _capi.MEAN):
        algorithm = int(algorithm) | st.hub_hints()  # the Storage knows its longest row / column: launches without the hub role
    if not (torch.is_grad_enabled() and (dense.requires_grad or (sparse.has_value and values.requires_grad))):
        # inference: nothing to record, skip the autograd.Function (~5 us; on the Cora/Pubmed class of graphs the whole
        # call is ~10 us, so that is a third of it): straight to the C ABI, or to the raw op when a plan exists
        if plan is None:
            return _capi.spmm(code, st.rowptr(), st.col(), values if sparse.has_value else None, dense, algorithm)[0]
    # what the Storage keeps next to its CSC view: values in CSC order (only the dense gradient's SpMM needs them) and
    # the plan of the transposed product
    need_d = torch.is_grad_enabled() and dense.requires_grad
    tvalues = st.csc_values() if (cuda and need_d and sparse.has_value and code in (_capi.SUM, _capi.MEAN)) else None
    plan_t, pinfo_t = st.spmm_plan('csc', dense.shape[1]) if (cuda and need_d and code in (_capi.SUM, _capi.MEAN)) \
        else (None, None)
    return op(st.rowptr(), st.col(), values, st.colptr(), st.csc_row(), st.csr2csc(), dense, sparse.has_value, algorithm,
              tvalues, plan, pinfo, plan_t, pinfo_t)


def spmm_sum(sparse: SparseTensor, dense: torch.Tensor, algorithm=0) -> torch.Tensor:
    r"""Sparse @ dense with sum reduction (algorithm is a tuning hint; all values give the same result)."""
    return _call(torch.ops.dgsparse_spmm.spmm_sum_p, _capi.SUM, sparse, dense, algorithm)


def spmm_mean(sparse: SparseTensor, dense: torch.Tensor, algorithm=0) -> torch.Tensor:
    r"""Sparse @ dense with mean reduction over each row's stored entries."""
    return _call(torch.ops.dgsparse_spmm.spmm_mean_p, _capi.MEAN, sparse, dense, algorithm)


def spmm_max(sparse: SparseTensor, dense: torch.Tensor, algorithm=0) -> torch.Tensor:
    r"""Row-wise max of val * dense[col]; empty rows give 0."""
    return _call(torch.ops.dgsparse_spmm.spmm_max_p, _capi.MAX, sparse, dense, algorithm)


def spmm_min(sparse: SparseTensor, dense: torch.Tensor, algorithm=0) -> torch.Tensor:
    r"""Row-wise min of val * dense[col]; empty rows give 0."""
    return _call(torch.ops.dgsparse_spmm.spmm_min_p, _capi.MIN, sparse, dense, algorithm)
