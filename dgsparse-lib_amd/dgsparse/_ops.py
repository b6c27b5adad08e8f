"""``torch.ops.dgsparse_spmm.*`` registry + the four autograd Functions, on top of the C ABI.

Mirrors the reference binding src/spmm.cpp:36-270 (TORCH_LIBRARY(dgsparse_spmm) with spmm_sum / spmm_max /
spmm_min / spmm_mean / csr2csc and torch::autograd::Function classes SpMMSum/Max/Min/Mean): same op names,
same positional schema ``(rowptr, col, values, colptr, row, csr2csc, dense, has_value, algorithm)``.
Backward = one SDDMM (grad of the sparse values) + one SpMM on the CSC arrays (grad of the dense operand),
as src/spmm.cpp:52-80,113-141, with three deliberate fixes:
  * mean backward uses the mathematically correct 1/deg(source row) weights (the reference divides by the
    column degree, src/spmm.cpp:224-253);
  * the dense gradient is only computed when needed (the reference dereferences grad_mat[0] unconditionally);
  * the dense gradient always has dense's shape, also when trailing columns of A are empty.
An extra op ``dgsparse_spmm::sddmm`` exposes SDDMM directly.
"""
import importlib.machinery
import os
import weakref

import torch

from . import _capi
from ._capi import MAX, MEAN, MIN, SUM

# Preferred binding: the C++ TORCH_LIBRARY in _spmm_hip*.so next to the package, found and loaded exactly like the
# reference does with _spmm_cuda*.so (dgsparse/__init__.py:16-26).  It calls the same C ABI; the Python registration
# below is the same binding written in Python (kept so that the op surface exists even where only the kernel library
# was built; force it with DGSPARSE_PY_BINDING=1).  Neither is a compute fallback: both end in libdgsparse_hip.so.
NATIVE_BINDING = False
_spec = importlib.machinery.PathFinder().find_spec('_spmm_hip', [os.path.dirname(__file__)])
if _spec is not None and os.environ.get('DGSPARSE_PY_BINDING', '0') != '1':
    torch.ops.load_library(_spec.origin)
    NATIVE_BINDING = True


_TV_CACHE = {}  # last `values[csr2csc]`: (weakref to the values tensor, its version, permutation address) -> tensor


def _t_values(values, csr2csc, has_value):
    """Edge values in CSC order.  The last result is kept (same rule as csrc/torch_binding.cpp: same tensor OBJECT via a
    weak reference, unchanged version counter, same permutation; never while a stream is being captured), because all
    layers that share an adjacency - and, with fixed weights, all iterations - ask for the same permuted values."""
    if not has_value:
        return None
    if not (csr2csc.dtype == torch.int32 and csr2csc.is_cuda and values.dtype == torch.float32):
        return values.view(-1).index_select(0, csr2csc.long() if csr2csc.dtype != torch.int64 else csr2csc)
    capturing = torch.cuda.is_current_stream_capturing()
    hit = _TV_CACHE.get('last')
    if (not capturing and hit is not None and hit[0]() is values and hit[1] == values._version
            and hit[2] == csr2csc.data_ptr() and hit[3].numel() == csr2csc.numel()):
        return hit[3]
    # one pass of the HIP gather over the int32 permutation (index_select wants an int64 copy first)
    out = _capi.gather_rows(values.detach().reshape(-1, 1), csr2csc).view(-1)
    if not capturing:
        _TV_CACHE['last'] = (weakref.ref(values), values._version, csr2csc.data_ptr(), out)
    return out


def _pad_rows(g, n):
    if g.shape[0] == n:
        return g
    out = g.new_zeros((n, g.shape[1]))
    out[:g.shape[0]] = g
    return out


class SpMMSum(torch.autograd.Function):
    """src/spmm.cpp:36-81"""

    @staticmethod
    def forward(ctx, rowptr, col, values, colptr, row, csr2csc, dense, has_value, algorithm):
        out, _ = _capi.spmm(SUM, rowptr, col, values if has_value else None, dense, algorithm)
        ctx.has_value, ctx.algorithm = has_value, algorithm
        ctx.save_for_backward(rowptr, col, values, colptr, row, csr2csc, dense)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        rowptr, col, values, colptr, row, csr2csc, dense = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        grad_value = grad_dense = None
        if ctx.has_value and ctx.needs_input_grad[2]:
            grad_value = _capi.sddmm(rowptr, col, grad_out, dense, SUM).view_as(values)
        if ctx.needs_input_grad[6]:
            g, _ = _capi.spmm(SUM, colptr, row, _t_values(values, csr2csc, ctx.has_value), grad_out, ctx.algorithm)
            grad_dense = _pad_rows(g, dense.shape[0])
        return None, None, grad_value, None, None, None, grad_dense, None, None


class SpMMMean(torch.autograd.Function):
    """src/spmm.cpp:208-262 (forward); backward with per-source-row 1/deg weights."""

    @staticmethod
    def forward(ctx, rowptr, col, values, colptr, row, csr2csc, dense, has_value, algorithm):
        out, _ = _capi.spmm(MEAN, rowptr, col, values if has_value else None, dense, algorithm)
        ctx.has_value, ctx.algorithm = has_value, algorithm
        ctx.save_for_backward(rowptr, col, values, colptr, row, csr2csc, dense)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        rowptr, col, values, colptr, row, csr2csc, dense = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        grad_value = grad_dense = None
        if ctx.has_value and ctx.needs_input_grad[2]:
            grad_value = _capi.sddmm(rowptr, col, grad_out, dense, MEAN).view_as(values)
        if ctx.needs_input_grad[6]:
            # d/dX mean_r = A^T diag(1/deg) dC: scale grad rows by 1/deg once, then a plain transposed SpMM
            deg = (rowptr[1:] - rowptr[:-1]).clamp_(min=1).to(torch.float32)
            g, _ = _capi.spmm(SUM, colptr, row, _t_values(values, csr2csc, ctx.has_value), grad_out / deg[:, None],
                              ctx.algorithm)
            grad_dense = _pad_rows(g, dense.shape[0])
        return None, None, grad_value, None, None, None, grad_dense, None, None


class _SpMMArg(torch.autograd.Function):
    """src/spmm.cpp:96-206 (SpMMMax / SpMMMin share everything but the reduce op)."""
    OP = MAX

    @classmethod
    def _fwd(cls, ctx, rowptr, col, values, colptr, row, csr2csc, dense, has_value, algorithm):
        out, E = _capi.spmm(cls.OP, rowptr, col, values if has_value else None, dense, algorithm)
        ctx.has_value, ctx.algorithm = has_value, algorithm
        ctx.save_for_backward(rowptr, col, values, colptr, row, csr2csc, dense, E)
        return out

    @staticmethod
    def _bwd(ctx, grad_out):
        rowptr, col, values, colptr, row, csr2csc, dense, E = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        grad_value = grad_dense = None
        need_v, need_d = ctx.has_value and ctx.needs_input_grad[2], ctx.needs_input_grad[6]
        if not torch.are_deterministic_algorithms_enabled() and (need_v or need_d):
            # one pass over the arg ids with fp32 atomics (csrc/arg_backward.hip); the masked kernels below are the
            # bit-reproducible route and serve torch.use_deterministic_algorithms(True)
            grad_dense, gw = _capi.spmm_arg_backward(rowptr, col, values if ctx.has_value else None, E, grad_out,
                                                     dense, need_dense=need_d, need_values=need_v)
            if need_v:
                grad_value = gw.view_as(values)
            return None, None, grad_value, None, None, None, grad_dense, None, None
        if ctx.has_value and ctx.needs_input_grad[2]:
            grad_value = _capi.sddmm(rowptr, col, grad_out, dense, SUM, E=E).view_as(values)
        if ctx.needs_input_grad[6]:
            grad_dense = _capi.spmm_mask(colptr, row, _t_values(values, csr2csc, ctx.has_value), grad_out, E,
                                         n_out=dense.shape[0])
        return None, None, grad_value, None, None, None, grad_dense, None, None


class SpMMMax(_SpMMArg):
    OP = MAX

    @staticmethod
    def forward(ctx, *a):
        return SpMMMax._fwd(ctx, *a)

    @staticmethod
    def backward(ctx, g):
        return _SpMMArg._bwd(ctx, g)


class SpMMMin(_SpMMArg):
    OP = MIN

    @staticmethod
    def forward(ctx, *a):
        return SpMMMin._fwd(ctx, *a)

    @staticmethod
    def backward(ctx, g):
        return _SpMMArg._bwd(ctx, g)


def _csr2csc_op(rowptr, colind, values):
    """src/spmm.cpp:91-94 -> [colptr, row, values in CSC order]; square like the reference (n = rows)."""
    n = rowptr.numel() - 1
    colptr, row, cscval, _ = _capi.csr2csc(rowptr, colind, values, n, want_perm=False)
    return [colptr, row, cscval]


if not NATIVE_BINDING:
    _SCHEMA = ('(Tensor rowptr, Tensor col, Tensor values, Tensor colptr, Tensor row, Tensor csr2csc, Tensor dense, '
               'bool has_value, int algorithm) -> Tensor')
    _lib = torch.library.Library('dgsparse_spmm', 'DEF')
    for _name, _fn in (('spmm_sum', SpMMSum), ('spmm_max', SpMMMax), ('spmm_min', SpMMMin), ('spmm_mean', SpMMMean)):
        _lib.define(_name + _SCHEMA)
        _lib.impl(_name, _fn.apply, 'CompositeImplicitAutograd')
    _lib.define('csr2csc(Tensor rowptr, Tensor colind, Tensor values) -> Tensor[]')
    _lib.impl('csr2csc', _csr2csc_op, 'CompositeImplicitAutograd')
    _lib.define('sddmm(Tensor rowptr, Tensor col, Tensor D1, Tensor D2, int reduce_op) -> Tensor')
    _lib.impl('sddmm', lambda rowptr, col, D1, D2, reduce_op: _capi.sddmm(rowptr, col, D1, D2, reduce_op),
              'CompositeImplicitAutograd')
    _lib.define('csr2csc_perm(Tensor rowptr, Tensor colind, int n_cols) -> Tensor[]')
    _lib.impl('csr2csc_perm', lambda rowptr, colind, n_cols: [t for i, t in enumerate(
        _capi.csr2csc(rowptr, colind, None, n_cols, want_perm=True)) if i != 2], 'CompositeImplicitAutograd')
    _lib.define('spmm_raw(int op, Tensor rowptr, Tensor col, Tensor values, Tensor dense, bool has_value, '
                'int algorithm) -> Tensor[]')
    _lib.impl('spmm_raw', lambda op, rowptr, col, values, dense, has_value, algorithm: [
        t for t in _capi.spmm(op, rowptr, col, values if has_value else None, dense, algorithm) if t is not None],
        'CompositeImplicitAutograd')
