"""``SparseTensor``: thin container over ``Storage`` -- mirrors the reference dgsparse/tensor.py:7-42."""
from typing import Optional

import torch

from .storage import Storage


class SparseTensor(object):
    storage: Storage

    def __init__(
        self,
        row: Optional[torch.Tensor] = None,
        rowptr: Optional[torch.Tensor] = None,
        col: Optional[torch.Tensor] = None,
        values: Optional[torch.Tensor] = None,
        has_value: bool = False,
    ):
        self.storage = Storage(row=row, rowptr=rowptr, col=col, values=values)
        self.has_value = has_value

    @classmethod
    def from_torch_sparse_csr_tensor(self, mat: torch.Tensor, has_value: bool = True, requires_grad: bool = False):
        """tensor.py:25-42: takes crow_indices / col_indices / values as they are (must already be int32)."""
        if has_value:
            values = mat.values()
            if requires_grad:
                values.requires_grad_()
        else:
            values = None
        return SparseTensor(row=None, rowptr=mat.crow_indices(), col=mat.col_indices(), values=values,
                            has_value=has_value)

    @property
    def sparse_sizes(self):
        return self.storage.sparse_sizes

    @property
    def nnz(self):
        return self.storage.nnz
