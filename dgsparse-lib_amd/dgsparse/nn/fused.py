"""``spmm_sum_fused``: SpMM-sum with the epilogue fused into the row-end store of the HIP kernels,

    out = relu(row_scale[:, None] * (A @ X) + bias)          every part optional

bit-identical to ``torch.relu(row_scale[:, None] * dgsparse.spmm_sum(A, X) + bias)`` but with the M x N result written
once instead of written, read and written again (``dgs_spmm_csr_ex_f32``).  New: the reference's GCN layer runs
``spmm_sum`` and ``torch.relu`` as two passes (dgsparse/nn/gcnconv.py:10-35).  Differentiable w.r.t. ``X``, the edge values
and ``bias``; ``row_scale`` is a constant (its gradient would need the pre-scale product this op never materialises)."""
import torch

from .. import _capi
from ..tensor import SparseTensor


class _SpMMSumFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sparse: SparseTensor, dense, values, bias, row_scale, relu: bool):
        st = sparse.storage
        has_value = sparse.has_value
        plan = None
        buf, info = st.spmm_plan('csr', dense.shape[1])
        if buf is not None:
            plan = _plan_obj(buf, info, st, 'csr')
        out, _ = _capi.spmm(_capi.SUM, st.rowptr(), st.col(), values.detach() if has_value else None, dense.detach(),
                            algorithm=st.hub_hints() & _capi.ALG_NO_HUB_ROWS, plan=plan, bias=None if bias is None else bias.detach(), row_scale=row_scale, relu=relu)
        ctx.sparse, ctx.relu, ctx.has_value = sparse, relu, has_value
        ctx.save_for_backward(dense, values, row_scale, out if relu else None)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, grad):
        dense, values, row_scale, out = ctx.saved_tensors
        st = ctx.sparse.storage
        g = grad.contiguous()
        if ctx.relu:
            g = g * (out > 0)
        g_bias = g.sum(0) if (ctx.has_bias and ctx.needs_input_grad[3]) else None
        if row_scale is not None:
            g = g * row_scale[:, None]
        g_dense = g_val = None
        N = dense.shape[1]
        if ctx.needs_input_grad[1]:
            buf, info = st.spmm_plan('csc', N)
            plan_t = _plan_obj(buf, info, st, 'csc') if buf is not None else None
            tv = st.csc_values() if ctx.has_value else None
            g_dense, _ = _capi.spmm(_capi.SUM, st.colptr(), st.csc_row(), tv, g, plan=plan_t,
                                    algorithm=_capi.ALG_NO_HUB_ROWS if st.hub_hints() & _capi.ALG_NO_HUB_COLS else 0)
            if g_dense.shape[0] < dense.shape[0]:
                g_dense = torch.cat([g_dense, g_dense.new_zeros((dense.shape[0] - g_dense.shape[0], N))])
        if ctx.has_value and ctx.needs_input_grad[2]:
            buf, info = st.spmm_plan('csr', N)
            plan = _plan_obj(buf, info, st, 'csr') if buf is not None else None
            g_val = _capi.sddmm(st.rowptr(), st.col(), g, dense.detach().contiguous(), plan=plan).view_as(values)
        return None, g_dense, g_val, g_bias, None, None


def _plan_obj(buf, info, st, which):
    """The Storage hands plans out as (device buffer, 16-int32 CPU tensor); the ctypes layer wants its SpmmPlan."""
    import ctypes
    pi = _capi.PlanInfo()
    ctypes.memmove(ctypes.byref(pi), info.data_ptr(), ctypes.sizeof(pi))
    if which == 'csr':
        ptr, idx, M, K = st.rowptr(), st.col(), st.sparse_sizes[0], st.sparse_sizes[1]
    else:
        ptr, idx, M, K = st.colptr(), st.csc_row(), st.sparse_sizes[1], st.sparse_sizes[0]
    return _capi.SpmmPlan(buf, pi, M, K, st.nnz, ptr.data_ptr(), idx.data_ptr())


def spmm_sum_fused(sparse: SparseTensor, dense: torch.Tensor, bias=None, row_scale=None, relu: bool = False) -> torch.Tensor:
    """relu(row_scale[:, None] * (sparse @ dense) + bias) in one pass over the output (see the module docstring)."""
    st = sparse.storage
    if dense.dim() != 2 or dense.shape[0] < st.sparse_sizes[1]:
        raise ValueError(f'dgsparse: dense has shape {tuple(dense.shape)} but the sparse tensor references '
                         f'{st.sparse_sizes[1]} columns')
    if row_scale is not None and row_scale.requires_grad:
        raise ValueError('dgsparse: row_scale of the fused epilogue is a constant (no gradient)')
    return _SpMMSumFused.apply(sparse, dense, st.values(), bias, row_scale, bool(relu))
