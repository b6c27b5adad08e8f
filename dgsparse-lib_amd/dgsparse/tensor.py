"""``SparseTensor``: what the operators take -- a ``Storage`` plus the ``has_value`` switch.

Same constructor and ``from_torch_sparse_csr_tensor`` factory as the reference's ``dgsparse/tensor.py:7-42``."""
from typing import Optional

import torch

from .storage import Storage


class SparseTensor(object):
    storage: Storage

    def __init__(self, row: Optional[torch.Tensor] = None, rowptr: Optional[torch.Tensor] = None,
                 col: Optional[torch.Tensor] = None, values: Optional[torch.Tensor] = None, has_value: bool = False):
        self.has_value = has_value
        self.storage = Storage(row=row, rowptr=rowptr, col=col, values=values)

    @classmethod
    def from_torch_sparse_csr_tensor(cls, mat: torch.Tensor, has_value: bool = True, requires_grad: bool = False):
        """Wraps a ``torch.sparse_csr_tensor`` without copying; its index tensors must already be int32, exactly as
        the reference requires (the Storage asserts fire otherwise).  With ``has_value`` the values tensor is shared
        and, if asked, marked as requiring grad; without it every stored entry weighs 1."""
        weights = None
        if has_value:
            weights = mat.values()
            if requires_grad:
                weights.requires_grad_()
        return cls(rowptr=mat.crow_indices(), col=mat.col_indices(), values=weights, has_value=has_value)

    @property
    def sparse_sizes(self):
        return self.storage.sparse_sizes

    @property
    def nnz(self):
        return self.storage.nnz
