"""Format transforms on a ``SparseTensor`` (reference ``dgsparse/ftransform.py:6-10``)."""
from typing import Tuple

import torch

from .tensor import SparseTensor


def csr2csc(sparse: SparseTensor) -> Tuple[torch.Tensor]:
    """(colptr, row indices, values) of the transposed matrix, values carried along in CSC order.

    The reference calls its square-only op ``dgsparse_spmm::csr2csc(rowptr, col, values)`` here; the Storage already
    holds the exact CSC view with the true column count (rectangular matrices included), so that is what is returned."""
    st = sparse.storage
    if st.nnz == 0 or not st.col().is_cuda:
        return st.colptr(), st.csc_row(), st.values()[:0] if st.nnz == 0 else st.values()[st.csr2csc().long()]
    return st.colptr(), st.csc_row(), st.csc_values()
