"""Format transforms on a ``SparseTensor`` (reference ``dgsparse/ftransform.py:6-10``)."""
from typing import Tuple

import torch

from .tensor import SparseTensor


def csr2csc(sparse: SparseTensor) -> Tuple[torch.Tensor]:
    """(colptr, row indices, values) of the transposed matrix, values carried along in CSC order."""
    st = sparse.storage
    return torch.ops.dgsparse_spmm.csr2csc(st.rowptr(), st.col(), st.values())
