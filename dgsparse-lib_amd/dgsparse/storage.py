"""``Storage`` -- the CSR arrays of a sparse matrix plus its CSC view.

Behavioural mirror of the reference's ``dgsparse/storage.py:6-174`` (constructor arguments, what is asserted, which
accessor raises ``ValueError`` when its array is absent, ``Storage.empty()``, eager CSR->CSC conversion), written
independently.  Two intended differences:

* the CSR->CSC permutation comes from the integer HIP ``csr2csc`` and is exact for any nnz -- the reference pushes
  ``arange(nnz)`` through cuSPARSE as float32 *values* (storage.py:164-169), which is exact only below 2**24 entries;
* rectangular matrices work: the CSC view has ``sparse_sizes[1] = col.max() + 1`` columns (the reference passes
  ``n, n`` to cuSPARSE and is square-only).
"""
from typing import Optional

import torch

from . import _capi

_INDEX = torch.int32


def _index_array(t: torch.Tensor, like: torch.Tensor, numel: Optional[int] = None, dtypes=(_INDEX,)) -> torch.Tensor:
    """Asserts the reference's invariants for an index array (dtype, 1-D, same device, length) -> contiguous."""
    assert t.dtype in dtypes
    assert t.dim() == 1
    assert t.device == like.device
    if numel is not None:
        assert t.numel() == numel
    return t.contiguous()


class Storage(object):
    # attribute names are part of the surface: the reference's tests and layers reach into them
    _row: Optional[torch.Tensor]
    _rowptr: Optional[torch.Tensor]
    _col: Optional[torch.Tensor]
    _values: Optional[torch.Tensor]
    _colptr: Optional[torch.Tensor]
    _csr2csc: Optional[torch.Tensor]
    _csc2csr: Optional[torch.Tensor]
    _colcount: Optional[torch.Tensor]

    def __init__(self, row=None, rowptr=None, col=None, values=None, colptr=None, csr2csc=None, csc2csr=None,
                 colcount=None):
        assert col is not None and (rowptr is not None or row is not None)
        assert col.dtype == _INDEX and col.dim() == 1
        col = col.contiguous()
        nnz = col.numel()

        if rowptr is not None:
            n_rows = rowptr.numel() - 1
        else:
            n_rows = int(row.max()) + 1 if row.numel() else 0
        n_cols = int(col.max()) + 1 if nnz else 0  # one device sync per construction, like the reference
        self.sparse_sizes = (n_rows, n_cols)
        self.nnz = nnz

        if row is not None:
            row = _index_array(row, col, nnz)
        if rowptr is not None:
            rowptr = _index_array(rowptr, col, n_rows + 1)
        else:  # COO rows (sorted, as CSR order requires) -> row pointer
            rowptr = torch.zeros(n_rows + 1, dtype=_INDEX, device=col.device)
            if nnz:
                rowptr[1:] = torch.cumsum(torch.bincount(row.long(), minlength=n_rows), 0)

        if values is None:  # unit weights when the caller has none (reference: torch.ones)
            values = torch.ones(nnz, dtype=torch.float32, device=col.device)
        else:
            assert values.device == col.device and values.size(0) == nnz
            values = values.contiguous()

        # optional pre-computed CSC pieces (the reference wants int64 here and narrows later; both are accepted)
        if colptr is not None:
            colptr = _index_array(colptr, col, n_cols + 1, (torch.int64, _INDEX)).to(_INDEX)
        if csr2csc is not None:
            csr2csc = _index_array(csr2csc, col, nnz, (torch.int64, _INDEX)).to(_INDEX)
        if colcount is not None:
            colcount = _index_array(colcount, col, n_cols, (torch.int64,))

        self._row, self._rowptr, self._col, self._values = row, rowptr, col, values
        self._colptr, self._csr2csc, self._csc2csr, self._colcount = colptr, csr2csc, csc2csr, colcount
        self.csr2csc_convert()

    @classmethod
    def empty(cls):
        """A 0 x 0 matrix on the CPU (reference storage.py:102-116)."""
        nothing = torch.tensor([], dtype=_INDEX)
        return cls(row=nothing, rowptr=None, col=nothing.clone())

    # ---- accessors: the array, or ValueError when it does not exist (reference storage.py:118-157) ----------------
    def _present(self, name: str) -> torch.Tensor:
        t = getattr(self, name)
        if t is None:
            raise ValueError
        return t

    def row(self) -> torch.Tensor:
        return self._present('_row')

    def rowptr(self) -> torch.Tensor:
        return self._present('_rowptr')

    def col(self) -> torch.Tensor:
        return self._present('_col')

    def colptr(self) -> torch.Tensor:
        return self._present('_colptr')

    def values(self) -> torch.Tensor:
        return self._present('_values')

    def csr2csc(self) -> torch.Tensor:
        return self._present('_csr2csc')

    def csr2csc_convert(self):
        """Fills in whatever is missing of (colptr, CSC row indices, CSR->CSC permutation), once.

        As in the reference (storage.py:170-173) the CSC row indices land in ``_row`` when no COO rows were given, and
        that is what the spmm operators pass as their ``row`` argument."""
        if None not in (self._csr2csc, self._colptr, self._row):
            return self._csr2csc
        if self.nnz == 0:
            dev = self._col.device
            colptr = torch.zeros(self.sparse_sizes[1] + 1, dtype=_INDEX, device=dev)
            csc_row = perm = torch.zeros(0, dtype=_INDEX, device=dev)
        else:
            colptr, csc_row, _, perm = _capi.csr2csc(self._rowptr, self._col, None, self.sparse_sizes[1], want_perm=True)
        if self._row is None:
            self._row = csc_row
        if self._colptr is None:
            self._colptr = colptr
        self._csr2csc = perm
        return perm
