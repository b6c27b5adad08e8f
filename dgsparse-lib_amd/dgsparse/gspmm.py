"""Generalised SpMM ``u (.) e`` -- the surface of the reference's ``gspmm-fp`` demo module (src/gspmm-fp/gspmm.cc:30-47:
pybind module ``spmm`` with ``GSpMM_u_e``, ``GSpMM_u`` and the enums ``REDUCEOP`` / ``COMPUTEOP``)."""
import enum

import torch

from . import _capi


class REDUCEOP(enum.IntEnum):  # src/gspmm-fp/gspmm.h:15
    SUM = 0
    MAX = 1
    MIN = 2
    MEAN = 3


class COMPUTEOP(enum.IntEnum):  # src/gspmm-fp/gspmm.h:16 -- compute(e, u): ADD e+u, SUB u-e, MUL e*u, DIV u/e
    ADD = 0
    SUB = 1
    MUL = 2
    DIV = 3


def GSpMM_u_e(A_rowptr: torch.Tensor, A_colind: torch.Tensor, A_csrVal: torch.Tensor, B: torch.Tensor,
              re_op: REDUCEOP, comp_op: COMPUTEOP) -> torch.Tensor:
    """out[r,:] = reduce over the row's entries of compute(A_csrVal[p], B[A_colind[p],:]) (gspmm.cc:9-18)."""
    return _capi.gspmm(int(re_op), int(comp_op), A_rowptr, A_colind, A_csrVal.reshape(-1), B)


def GSpMM_u(A_rowptr: torch.Tensor, A_colind: torch.Tensor, B: torch.Tensor, op: REDUCEOP) -> torch.Tensor:
    """Copy-u aggregation without edge values (gspmm.cc:20-28): out[r,:] = reduce B[A_colind[p],:]."""
    return _capi.gspmm(int(op), int(COMPUTEOP.MUL), A_rowptr, A_colind, None, B)
