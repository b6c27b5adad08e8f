"""CPU-side tests: the C-ABI library loads and exports every symbol include/dgsparse_hip.h declares, the
Python surface mirrors the reference's names, host-side validation fires, and nothing in the product
package touches oracle/.  No compute calls (there is no GPU here)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'dgsparse-lib_amd')


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'dgsparse_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(dgs_\w+|gespmmCsrSpMM|spmm_cuda\w*|sddmm_cuda_csr)\s*\(', hdr))
    assert len(declared) >= 16
    lib = ctypes.CDLL(os.path.join(PKG, 'dgsparse', 'libdgsparse_hip.so'))
    for sym in sorted(declared):
        assert hasattr(lib, sym), f'{sym} declared in include/dgsparse_hip.h but not exported'
    from dgsparse import _capi
    assert set(_capi.EXPORTS) == declared
    assert _capi.arch() == 'gfx950' and _capi.version() >= 1000
    lib.dgs_strerror.restype = ctypes.c_char_p
    assert lib.dgs_strerror(-2) == b'workspace too small'


def test_python_surface_matches_reference_names():
    import dgsparse
    for name in ['spmm_sum', 'spmm_max', 'spmm_min', 'spmm_mean', 'Storage', 'SparseTensor', 'csr2csc']:
        assert hasattr(dgsparse, name)  # reference dgsparse/__init__.py:46-49
    assert dgsparse._C.cuda_version() == -1
    for op in ['spmm_sum', 'spmm_max', 'spmm_min', 'spmm_mean', 'csr2csc', 'sddmm']:
        assert hasattr(torch.ops.dgsparse_spmm, op)
    schema = str(torch.ops.dgsparse_spmm.spmm_sum.default._schema)
    assert 'Tensor rowptr, Tensor col, Tensor values, Tensor colptr, Tensor row, Tensor csr2csc, Tensor dense, ' \
           'bool has_value, int algorithm' in schema


def test_no_cpu_fallback_and_validation():
    import dgsparse
    from dgsparse import _capi
    rp = torch.tensor([0, 1, 2], dtype=torch.int)
    col = torch.tensor([0, 1], dtype=torch.int)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _capi.spmm(0, rp, col, None, torch.rand(2, 4))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        dgsparse.SparseTensor(rowptr=rp, col=col)
    with pytest.raises(AssertionError):
        dgsparse.Storage(rowptr=rp.long(), col=col)  # reference storage.py:54
    with pytest.raises(AssertionError):
        dgsparse.Storage(rowptr=rp, col=col.long())  # reference storage.py:29
    st = dgsparse.Storage.empty()  # reference storage.py:103
    assert st.sparse_sizes == (0, 0) and st.nnz == 0
    # pre-supplied CSC arrays skip the conversion (storage.py:160-161), so host logic is testable on CPU
    st = dgsparse.Storage(rowptr=rp, col=col, row=torch.tensor([0, 1], dtype=torch.int),
                          colptr=torch.tensor([0, 1, 2]), csr2csc=torch.tensor([0, 1]))
    assert st.colptr().dtype == torch.int and st.csr2csc().tolist() == [0, 1]
    assert st.values().tolist() == [1.0, 1.0]  # defaults to ones (storage.py:60-68)
    st._colptr = None
    with pytest.raises(ValueError):
        st.colptr()


def test_product_package_never_touches_the_oracle():
    bad = []
    for dp, _, fs in os.walk(PKG):
        for f in fs:
            if f.endswith(('.py', '.hip', '.h', '.cpp', 'Makefile')):
                txt = open(os.path.join(dp, f), errors='ignore').read()
                if re.search(r'^\s*(import|from)\s+oracle\b', txt, flags=re.M) or 'liborc' in txt or 'oracle/' in txt:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
