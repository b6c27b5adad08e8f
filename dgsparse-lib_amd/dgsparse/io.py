"""MatrixMarket ingestion with the semantics of the reference's loader (example/util/sp_util.hpp:171-251,
``read_mtx_file``): coordinate files only; VALUES ARE DROPPED (pattern only); symmetric matrices are mirrored;
entries are sorted by (row, col); duplicates are removed only for symmetric inputs (as the reference does).
One reference quirk is NOT reproduced: for symmetric files its CSR loop is bounded by the file's entry count
instead of the mirrored count (sp_util.hpp:238-247), which silently drops the tail of the matrix; here the whole
mirrored matrix is kept (tests/test_host_cpu.py pins the reference output as a prefix of this one).
Returns numpy CSR arrays; ``to_sparse_tensor`` uploads them as a ``dgsparse.SparseTensor``."""
import numpy as np


def read_mtx(path: str):
    """-> (nrow, ncol, rowptr int32 [nrow+1], col int32 [nnz])"""
    with open(path, 'r') as f:
        banner = f.readline().strip().lower().split()
        if len(banner) < 5 or banner[0] != '%%matrixmarket' or banner[1] != 'matrix' or banner[2] != 'coordinate':
            raise ValueError(f'{path}: only "%%MatrixMarket matrix coordinate ..." files are supported')
        field, symmetry = banner[3], banner[4]
        line = f.readline()
        while line.startswith('%') or not line.strip():
            line = f.readline()
        nrow, ncol, nnz = (int(x) for x in line.split()[:3])
        data = np.loadtxt(f, dtype=np.float64, ndmin=2, max_rows=nnz)
    if data.shape[0] < nnz:
        raise ValueError(f'{path}: not enough entries ({data.shape[0]} < {nnz})')
    r = data[:, 0].astype(np.int64) - 1  # mtx is 1-based
    c = data[:, 1].astype(np.int64) - 1
    del field
    if symmetry == 'symmetric':
        r, c = np.concatenate([r, c]), np.concatenate([c, r])
        key = np.unique(r * ncol + c)  # sorted + deduplicated (sp_util.hpp:219-229)
    else:
        key = np.sort(r * ncol + c, kind='stable')  # sorted, duplicates kept (sp_util.hpp:230-232)
    r = key // ncol
    col = (key - r * ncol).astype(np.int32)
    rowptr = np.zeros(nrow + 1, np.int64)
    np.cumsum(np.bincount(r, minlength=nrow), out=rowptr[1:])
    return nrow, ncol, rowptr.astype(np.int32), col


def to_sparse_tensor(rowptr, col, values=None, device='cuda', has_value=True):
    import torch

    from .tensor import SparseTensor
    v = None if values is None else torch.as_tensor(values, dtype=torch.float32).to(device)
    return SparseTensor(row=None, rowptr=torch.as_tensor(rowptr, dtype=torch.int32).to(device),
                        col=torch.as_tensor(col, dtype=torch.int32).to(device), values=v, has_value=has_value)
