"""CPU parity oracle for the dgSPARSE CSR SpMM / SDDMM / csr2csc hot path.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py -- never from the product package (dgsparse-lib_amd/), which must fail loudly when its
HIP library is missing instead of falling back to anything here.

Two back ends, both reached through ctypes on numpy arrays:
  * liborc.so            -- dgs_oracle.c, the restatement of the reference kernels (citations there).
  * _ref/libdgsref.so    -- the reference's own spmm_reference_host / sddmm_reference_host /
                            read_mtx_file compiled in place from /root/reference (ref_shim.cpp);
                            present only where it was built (this container; travels as a binary).
Pinning status: pinned (see oracle/README.md and tests/test_oracle_pin.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SUM, MAX, MIN, MEAN = 0, 1, 2, 3
REDUCE = {'sum': SUM, 'max': MAX, 'min': MIN, 'mean': MEAN}

_i32p = ctypes.POINTER(ctypes.c_int32)
_f32p = ctypes.POINTER(ctypes.c_float)
_i64 = ctypes.c_int64


def build(ref: bool = True) -> None:
    """Compile liborc.so (always) and _ref/libdgsref.so (when /root/reference exists)."""
    subprocess.check_call(['make', '-s', '-C', _HERE, 'all'])
    if ref and os.path.isdir('/root/reference'):
        subprocess.check_call(['make', '-s', '-C', _HERE, 'ref'])


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, 'liborc.so')
        if not os.path.exists(path):
            build(ref=False)
        _lib = ctypes.CDLL(path)
        _lib.orc_spmm_csr_f32.argtypes = [ctypes.c_int, ctypes.c_int, _i64, _i64, _i64, _i64, _i32p,
                                          _i32p, _f32p, _f32p, _f32p, _i32p, ctypes.c_int]
        _lib.orc_spmm_csr_mask_f32.argtypes = [ctypes.c_int, _i64, _i64, _i32p, _i32p, _f32p, _f32p,
                                               _i32p, _f32p]
        _lib.orc_sddmm_csr_f32.argtypes = [ctypes.c_int, ctypes.c_int, _i64, _i64, _i64, _i32p,
                                           _i32p, _f32p, _f32p, _f32p, ctypes.c_int]
        _lib.orc_sddmm_csr_mask_f32.argtypes = [ctypes.c_int, _i64, _i64, _i64, _i32p, _i32p, _f32p,
                                                _f32p, _i32p, _f32p]
        _lib.orc_csr2csc_i32.argtypes = [_i64, _i64, _i64, _i32p, _i32p, _f32p, _i32p, _i32p, _f32p,
                                         _i32p]
    return _lib


def have_ref() -> bool:
    return os.path.exists(os.path.join(_HERE, '_ref', 'libdgsref.so'))


def ref():
    """The reference's own CPU loops (None if oracle/_ref was not built)."""
    global _ref
    if _ref is None and have_ref():
        _ref = ctypes.CDLL(os.path.join(_HERE, '_ref', 'libdgsref.so'))
        ci = ctypes.c_int
        _ref.ref_spmm_sum.argtypes = [ci, ci, ci, _i32p, _i32p, _f32p, _f32p, _f32p]
        _ref.ref_sddmm.argtypes = [ci, ci, ci, ci, _i32p, _i32p, _f32p, _f32p, _f32p]
        _ref.ref_read_mtx.argtypes = [ctypes.c_char_p, _i32p, _i32p, _i32p, _i32p, _i32p]
    return _ref


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


def _f(a):
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def spmm(reduce, rowptr, col, val, B, K=None, fma=False, threads=1):
    """Algorithm-0 SpMM.  Returns (C[M,N] f32, E[M,N] i32)."""
    op = REDUCE[reduce] if isinstance(reduce, str) else int(reduce)
    rowptr, prp = _i(rowptr)
    col, pc = _i(col)
    val, pv = _f(val)
    B, pb = _f(B)
    M, N = rowptr.shape[0] - 1, B.shape[1]
    K = B.shape[0] if K is None else K
    C = np.empty((M, N), np.float32)
    E = np.empty((M, N), np.int32)
    rc = lib().orc_spmm_csr_f32(op, int(fma), M, K, N, col.shape[0], prp, pc, pv, pb,
                                C.ctypes.data_as(_f32p), E.ctypes.data_as(_i32p), int(threads))
    assert rc == 0
    return C, E


def spmm_sum_f64(rowptr, col, val, B, mean=False, absval=False):
    """Exact-arithmetic yardstick (double accumulation) for sum/mean - see dgs_oracle.c."""
    rowptr, prp = _i(rowptr)
    col, pc = _i(col)
    val, pv = _f(val)
    B, pb = _f(B)
    M, N = rowptr.shape[0] - 1, B.shape[1]
    C = np.empty((M, N), np.float64)
    f = lib().orc_spmm_sum_f64
    f.argtypes = [ctypes.c_int, ctypes.c_int, _i64, _i64, _i32p, _i32p, _f32p, _f32p, ctypes.POINTER(ctypes.c_double)]
    rc = f(int(mean), int(absval), M, N, prp, pc, pv, pb, C.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    assert rc == 0
    return C


def gspmm(reduce, compute, rowptr, col, val, B):
    """Generalised SpMM (gspmm-fp): compute in {'add','sub','mul','div'} = COMPUTEOP {0,1,2,3}."""
    rop = REDUCE[reduce] if isinstance(reduce, str) else int(reduce)
    cop = {'add': 0, 'sub': 1, 'mul': 2, 'div': 3}[compute] if isinstance(compute, str) else int(compute)
    rowptr, p0 = _i(rowptr)
    col, p1 = _i(col)
    val, p2 = _f(val)
    B, p3 = _f(B)
    C = np.empty((rowptr.shape[0] - 1, B.shape[1]), np.float32)
    f = lib().orc_gspmm_csr_f32
    f.argtypes = [ctypes.c_int, ctypes.c_int, _i64, _i64, _i32p, _i32p, _f32p, _f32p, _f32p]
    assert f(rop, cop, rowptr.shape[0] - 1, B.shape[1], p0, p1, p2, p3, C.ctypes.data_as(_f32p)) == 0
    return C


def spmm_mask(colptr, row, tval, G, E, fma=False):
    """max/min backward w.r.t. dense, on the CSC arrays.  Returns gX[Kcols,N]."""
    colptr, p0 = _i(colptr)
    row, p1 = _i(row)
    tval, p2 = _f(tval)
    G, p3 = _f(G)
    E, p4 = _i(E)
    Mo, N = colptr.shape[0] - 1, G.shape[1]
    out = np.empty((Mo, N), np.float32)
    rc = lib().orc_spmm_csr_mask_f32(int(fma), Mo, N, p0, p1, p2, p3, p4, out.ctypes.data_as(_f32p))
    assert rc == 0
    return out


def spmm_mask_f64(colptr, row, tval, G, E, absval=False):
    """float64 yardstick of spmm_mask (absval: condition scale sum|val*G| over the passing entries)."""
    colptr, p0 = _i(colptr)
    row, p1 = _i(row)
    tval, p2 = _f(tval)
    G, p3 = _f(G)
    E, p4 = _i(E)
    Mo, N = colptr.shape[0] - 1, G.shape[1]
    out = np.empty((Mo, N), np.float64)
    f = lib().orc_spmm_mask_f64
    f.argtypes = [ctypes.c_int, _i64, _i64, _i32p, _i32p, _f32p, _f32p, _i32p, ctypes.POINTER(ctypes.c_double)]
    assert f(int(absval), Mo, N, p0, p1, p2, p3, p4, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == 0
    return out


def sddmm(rowptr, col, D1, D2, reduce='sum', fma=False, threads=1):
    op = REDUCE[reduce] if isinstance(reduce, str) else int(reduce)
    rowptr, p0 = _i(rowptr)
    col, p1 = _i(col)
    D1, p2 = _f(D1)
    D2, p3 = _f(D2)
    out = np.empty(col.shape[0], np.float32)
    rc = lib().orc_sddmm_csr_f32(op, int(fma), rowptr.shape[0] - 1, D1.shape[1], col.shape[0], p0, p1,
                                 p2, p3, out.ctypes.data_as(_f32p), int(threads))
    assert rc == 0
    return out


def sddmm_mask(rowptr, col, D1, D2, E, fma=False):
    rowptr, p0 = _i(rowptr)
    col, p1 = _i(col)
    D1, p2 = _f(D1)
    D2, p3 = _f(D2)
    E, p4 = _i(E)
    out = np.empty(col.shape[0], np.float32)
    rc = lib().orc_sddmm_csr_mask_f32(int(fma), rowptr.shape[0] - 1, D1.shape[1], col.shape[0], p0,
                                      p1, p2, p3, p4, out.ctypes.data_as(_f32p))
    assert rc == 0
    return out


def csr2csc(rowptr, col, val, Kcols):
    """Stable transpose.  Returns (colptr[K+1], row[nnz], cscval[nnz] | None, perm[nnz])."""
    rowptr, p0 = _i(rowptr)
    col, p1 = _i(col)
    val, p2 = _f(val)
    nnz = col.shape[0]
    colptr = np.empty(Kcols + 1, np.int32)
    row = np.empty(nnz, np.int32)
    perm = np.empty(nnz, np.int32)
    cscval = np.empty(nnz, np.float32) if val is not None else None
    rc = lib().orc_csr2csc_i32(rowptr.shape[0] - 1, Kcols, nnz, p0, p1, p2,
                               colptr.ctypes.data_as(_i32p), row.ctypes.data_as(_i32p),
                               cscval.ctypes.data_as(_f32p) if cscval is not None else None,
                               perm.ctypes.data_as(_i32p))
    assert rc == 0, rc
    return colptr, row, cscval, perm


# ---- the reference's own loops (oracle/_ref) -------------------------------------------------
def ref_spmm_sum(rowptr, col, val, B):
    r = ref()
    assert r is not None, 'oracle/_ref not built'
    rowptr, p0 = _i(rowptr)
    col, p1 = _i(col)
    val, p2 = _f(val if val is not None else np.ones(col.shape[0], np.float32))
    B, p3 = _f(B)
    M, N = rowptr.shape[0] - 1, B.shape[1]
    C = np.empty((M, N), np.float32)
    r.ref_spmm_sum(M, N, B.shape[0], p0, p1, p2, p3, C.ctypes.data_as(_f32p))
    return C


def ref_sddmm(rowptr, col, D1, D2):
    r = ref()
    assert r is not None, 'oracle/_ref not built'
    rowptr, p0 = _i(rowptr)
    col, p1 = _i(col)
    D1, p2 = _f(D1)
    D2, p3 = _f(D2)
    out = np.empty(col.shape[0], np.float32)
    r.ref_sddmm(rowptr.shape[0] - 1, D2.shape[0], D1.shape[1], col.shape[0], p0, p1, p2, p3,
                out.ctypes.data_as(_f32p))
    return out


def ref_read_mtx(path):
    """CSR pattern via the reference's read_mtx_file (values dropped, symmetrised, sorted)."""
    r = ref()
    assert r is not None, 'oracle/_ref not built'
    n = (ctypes.c_int32 * 3)()
    r.ref_read_mtx(path.encode(), ctypes.cast(ctypes.byref(n, 0), _i32p),
                   ctypes.cast(ctypes.byref(n, 4), _i32p), ctypes.cast(ctypes.byref(n, 8), _i32p),
                   None, None)
    nrow, ncol, nnz = n[0], n[1], n[2]
    ip = np.empty(nrow + 1, np.int32)
    ix = np.empty(nnz, np.int32)
    r.ref_read_mtx(path.encode(), None, None, None, ip.ctypes.data_as(_i32p),
                   ix.ctypes.data_as(_i32p))
    return nrow, ncol, ip, ix


def max_threads() -> int:
    return int(lib().orc_max_threads())
