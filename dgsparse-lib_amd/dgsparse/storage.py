"""``Storage``: CSR arrays + the eagerly built CSC view, mirroring the reference dgsparse/storage.py:6-174
(same constructor arguments, asserts, accessors raising ValueError, ``Storage.empty()``).

Differences (all inside the reference's documented intent):
  * the CSR->CSC permutation is computed in integers by the HIP csr2csc (exact for any nnz; the reference
    pushes arange(nnz) through cuSPARSE as float32 values, storage.py:164-169, exact only below 2^24);
  * rectangular matrices work: the CSC has ``sparse_sizes[1] = col.max()+1`` columns (reference: n x n only).
"""
from typing import Optional

import torch

from . import _capi


class Storage(object):
    _row: Optional[torch.Tensor]
    _rowptr: Optional[torch.Tensor]
    _col: Optional[torch.Tensor]
    _values: Optional[torch.Tensor]
    _colptr: torch.Tensor
    _csr2csc: torch.Tensor
    _csc2csr: torch.Tensor
    _colcount: Optional[torch.Tensor]

    def __init__(
        self,
        row: Optional[torch.Tensor] = None,
        rowptr: Optional[torch.Tensor] = None,
        col: Optional[torch.Tensor] = None,
        values: Optional[torch.Tensor] = None,
        colptr: Optional[torch.Tensor] = None,
        csr2csc: Optional[torch.Tensor] = None,
        csc2csr: Optional[torch.Tensor] = None,
        colcount: Optional[torch.Tensor] = None,
    ):
        assert row is not None or rowptr is not None
        assert col is not None
        assert col.dtype == torch.int
        assert col.dim() == 1
        col = col.contiguous()

        M: int = 0
        if rowptr is not None:
            M = rowptr.numel() - 1
        elif row is not None and row.numel() > 0:
            M = int(row.max()) + 1

        N: int = 0
        if col.numel() > 0:
            N = int(col.max()) + 1  # one device sync per construction, as the reference (storage.py:41-43)

        self.sparse_sizes = (M, N)
        self.nnz = col.size(0)

        if row is not None:
            assert row.dtype == torch.int
            assert row.device == col.device
            assert row.dim() == 1
            assert row.numel() == col.numel()
            row = row.contiguous()

        if rowptr is not None:
            assert rowptr.dtype == torch.int
            assert rowptr.device == col.device
            assert rowptr.dim() == 1
            assert rowptr.numel() - 1 == self.sparse_sizes[0]
            rowptr = rowptr.contiguous()

        if values is not None:
            assert values.device == col.device
            assert values.size(0) == self.nnz
            values = values.contiguous()
        else:
            values = torch.ones((self.nnz), dtype=torch.float, device=col.device)

        if colptr is not None:
            assert colptr.dtype in (torch.long, torch.int)
            assert colptr.device == col.device
            assert colptr.dim() == 1
            assert colptr.numel() - 1 == self.sparse_sizes[1]
            colptr = colptr.contiguous().to(torch.int)

        if csr2csc is not None:
            assert csr2csc.dtype in (torch.long, torch.int)
            assert csr2csc.device == col.device
            assert csr2csc.dim() == 1
            assert csr2csc.numel() == col.size(0)
            csr2csc = csr2csc.contiguous().to(torch.int)

        if colcount is not None:
            assert colcount.dtype == torch.long
            assert colcount.device == col.device
            assert colcount.dim() == 1
            assert colcount.numel() == self.sparse_sizes[1]
            colcount = colcount.contiguous()

        if rowptr is None:  # COO rows given: build rowptr (rows must be sorted, as CSR order requires)
            counts = torch.bincount(row.long(), minlength=M)
            rowptr = torch.zeros(M + 1, dtype=torch.int, device=col.device)
            rowptr[1:] = torch.cumsum(counts, 0)

        self._row = row
        self._rowptr = rowptr
        self._col = col
        self._values = values
        self._colptr = colptr
        self._csr2csc = csr2csc
        self._csc2csr = csc2csr
        self._colcount = colcount

        # convert
        self.csr2csc_convert()

    @classmethod
    def empty(self):
        row = torch.tensor([], dtype=torch.int)
        col = torch.tensor([], dtype=torch.int)
        return Storage(row=row, rowptr=None, col=col, values=None, colptr=None, csc2csr=None, csr2csc=None,
                       colcount=None)

    def row(self) -> torch.Tensor:
        row = self._row
        if row is not None:
            return row
        else:
            raise ValueError

    def rowptr(self) -> torch.Tensor:
        rowptr = self._rowptr
        if rowptr is not None:
            return rowptr
        else:
            raise ValueError

    def col(self) -> torch.Tensor:
        col = self._col
        if col is not None:
            return col
        else:
            raise ValueError

    def colptr(self) -> torch.Tensor:
        colptr = self._colptr
        if colptr is not None:
            return colptr
        else:
            raise ValueError

    def values(self) -> torch.Tensor:
        values = self._values
        if values is not None:
            return values
        else:
            raise ValueError

    def csr2csc(self) -> torch.Tensor:
        csr2csc = self._csr2csc
        if csr2csc is not None:
            return csr2csc
        else:
            raise ValueError

    def csr2csc_convert(self):
        """Builds (colptr, row-of-CSC, csr2csc permutation) once; storage.py:159-174 in the reference.

        NB the reference stores the CSC row indices in ``_row`` (storage.py:170-171) and spmm passes that as the
        ``row`` argument of the op; the same convention is kept."""
        if self._csr2csc is not None and self._colptr is not None and self._row is not None:
            return self._csr2csc
        if self.nnz == 0:
            dev = self._col.device
            self._colptr = torch.zeros(self.sparse_sizes[1] + 1, dtype=torch.int, device=dev)
            self._row = torch.zeros(0, dtype=torch.int, device=dev)
            self._csr2csc = torch.zeros(0, dtype=torch.int, device=dev)
            return self._csr2csc
        colptr, row, _, perm = _capi.csr2csc(self._rowptr, self._col, None, self.sparse_sizes[1], want_perm=True)
        self._row = row
        self._colptr = colptr
        self._csr2csc = perm
        return self._csr2csc
