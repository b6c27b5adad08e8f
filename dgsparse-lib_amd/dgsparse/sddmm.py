"""``dgsparse.sddmm`` -- public SDDMM entry (new: in the reference SDDMM is reachable only as the backward
of spmm_*, src/spmm.cpp:66,127,183,238, and through the standalone C library src/sddmm/sddmm.h:7-11)."""
import torch

from .tensor import SparseTensor


def sddmm(sparse: SparseTensor, D1: torch.Tensor, D2: torch.Tensor, reduce: str = 'sum') -> torch.Tensor:
    r"""out[e] = <D1[row(e)], D2[col(e)]> for every stored entry e (CSR order); ``reduce='mean'`` divides by the
    degree of row(e) like the reference's MEAN instantiation (include/cuda/sddmm_cuda.cuh:266-272)."""
    op = {'sum': 0, 'mean': 3}[reduce]
    st = sparse.storage
    return torch.ops.dgsparse_spmm.sddmm(st.rowptr(), st.col(), D1, D2, op)
