"""``csr2csc(sparse)`` -- mirrors the reference dgsparse/ftransform.py:6-10."""
from typing import Tuple

import torch

from .tensor import SparseTensor


def csr2csc(sparse: SparseTensor) -> Tuple[torch.Tensor]:
    rowptr = sparse.storage._rowptr
    col = sparse.storage._col
    values = sparse.storage._values
    return torch.ops.dgsparse_spmm.csr2csc(rowptr, col, values)
