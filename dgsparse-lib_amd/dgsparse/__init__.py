"""dgsparse on MI355X: the reference's Python operator surface (dgsparse/__init__.py:1-49) over hand-written
gfx950 HIP kernels reached through the C ABI of ``libdgsparse_hip.so`` (include/dgsparse_hip.h).

``import dgsparse`` fails with ImportError when the HIP library has not been built - there is no fallback.
"""
from . import _C  # noqa: F401  (reference: pybind module with cuda_version())
from . import _capi  # noqa: F401  (raises ImportError if libdgsparse_hip.so is missing)
from . import _ops  # noqa: F401  (registers torch.ops.dgsparse_spmm.*)
from .ftransform import csr2csc
from .sddmm import sddmm
from .spmm import spmm_max, spmm_mean, spmm_min, spmm_sum
from .storage import Storage
from .tensor import SparseTensor
from . import nn  # noqa: F401,E402
from . import gspmm  # noqa: F401,E402  (GSpMM_u_e / GSpMM_u of the reference's gspmm-fp module)

__version__ = '0.1'

# `algorithm` bits above the reference's algorithm ids (include/dgsparse_hip.h): dgsparse.spmm_sum(A, X, dgsparse.ALG_STRICT_SUM)
ALG_SHARED_GPU = _capi.ALG_SHARED_GPU
ALG_STRICT_SUM = _capi.ALG_STRICT_SUM      # sum / mean as ONE sequential fmaf chain per (row, feature), any row length
ALG_STRICT_NOFMA = _capi.ALG_STRICT_NOFMA  # ... with the product rounded before the add (the reference's host loop)
ALG_NO_HUB_ROWS = _capi.ALG_NO_HUB_ROWS    # the caller knows that no row is longer than the hub threshold (spmm_* add it themselves
ALG_NO_HUB_COLS = _capi.ALG_NO_HUB_COLS    # from what the Storage knows: Storage.hub_hints()); ... no column (backward's product)

cuda_version = _C.cuda_version()  # -1 on ROCm: the reference's CUDA-major check is skipped (__init__.py:29)

__all__ = ['spmm_sum', 'spmm_max', 'spmm_min', 'spmm_mean', 'sddmm', 'Storage', 'SparseTensor', 'csr2csc']
