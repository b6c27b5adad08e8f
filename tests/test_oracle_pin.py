"""Pins the CPU oracle (oracle/dgs_oracle.c) against vectors that come from OUTSIDE this repo:
torch.sparse.mm / _sparse_mm_reduce_impl (the reference tests' oracle), the reference's own
spmm_reference_host / sddmm_reference_host (oracle/_ref) and scipy tocsc -- see tests/golden/make_golden.py.
CPU only."""
import os

import numpy as np
import pytest

import oracle
from util import assert_bitexact, assert_close, golden_names, load_golden

CASES = golden_names()


@pytest.mark.parametrize('name', CASES)
def test_spmm_sum_bitexact_vs_reference_host_loop(name):
    g = load_golden(name)
    C, _ = oracle.spmm('sum', g['rowptr'], g['col'], g['val'], g['X'])
    assert_bitexact(C, g['ref_sum_out'], 'sum vs spmm_reference_host')  # sp_util.hpp:63-84


@pytest.mark.parametrize('name', CASES)
@pytest.mark.parametrize('reduce', ['sum', 'mean'])
def test_spmm_sum_mean_vs_torch(name, reduce):
    g = load_golden(name)
    C, _ = oracle.spmm(reduce, g['rowptr'], g['col'], g['val'], g['X'])
    # torch's CPU kernel accumulates in a different order -> tolerance (north_star: 1e-5 rel)
    assert_close(C, g[f'{reduce}_out'], rtol=1e-5, atol=2e-6, what=reduce)
    Cf, _ = oracle.spmm(reduce, g['rowptr'], g['col'], g['val'], g['X'], fma=True)
    assert_close(Cf, g[f'{reduce}_out'], rtol=1e-5, atol=2e-6, what=reduce + '+fma')


@pytest.mark.parametrize('name', CASES)
@pytest.mark.parametrize('reduce', ['max', 'min'])
def test_spmm_max_min_bitexact_vs_torch(name, reduce):
    g = load_golden(name)
    C, E = oracle.spmm(reduce, g['rowptr'], g['col'], g['val'], g['X'])
    assert_bitexact(C, g[f'{reduce}_out'], reduce + ' values')
    assert_bitexact(E, g[f'{reduce}_E'], reduce + ' arg column ids')


@pytest.mark.parametrize('name', CASES)
def test_sddmm_bitexact_vs_reference_host_loop(name):
    g = load_golden(name)
    out = oracle.sddmm(g['rowptr'], g['col'], g['D1'], g['X'])
    assert_bitexact(out, g['ref_sddmm_out'], 'sddmm vs sddmm_reference_host')  # sp_util.hpp:88-112


@pytest.mark.parametrize('name', CASES)
def test_csr2csc_vs_scipy(name):
    g = load_golden(name)
    colptr, row, cscval, perm = oracle.csr2csc(g['rowptr'], g['col'], g['val'], int(g['K']))
    assert_bitexact(colptr, g['csc_colptr'])
    assert_bitexact(row, g['csc_row'])
    assert_bitexact(cscval, g['csc_val'])
    assert_bitexact(perm, g['csc_perm'])


def test_csr2csc_reference_fixture_p2p_gnutella31():
    """The reference's own test (test/test_csr2csr.py) on its own data file, expected from scipy."""
    g = load_golden('p2p_gnutella31_csr2csc')
    colptr, row, cscval, _ = oracle.csr2csc(g['rowptr'], g['col'], g['val'], int(g['shape'][1]))
    assert_bitexact(colptr, g['csc_colptr'])
    assert_bitexact(row, g['csc_row'])
    assert_bitexact(cscval, g['csc_val'])


GRAD = [n for n in CASES if 'grad' in n or n in ('cora_shaped_N32', 'small_weighted_N64')]


@pytest.mark.parametrize('name', GRAD)
def test_backward_formulas_vs_torch_autograd(name):
    """dX = A^T-SpMM, dA = SDDMM (sum); masked variants (max); mean = rows scaled by 1/deg."""
    g = load_golden(name)
    rp, col, val, X, G = g['rowptr'], g['col'], g['val'], g['X'], g['G']
    K = int(g['K'])
    colptr, row, tval, perm = oracle.csr2csc(rp, col, val, K)
    # sum
    dX, _ = oracle.spmm('sum', colptr, row, tval, G)
    assert_close(dX, g['sum_dX'], rtol=1e-5, atol=2e-6, what='sum dX')
    assert_close(oracle.sddmm(rp, col, G, X), g['sum_dA'], rtol=1e-5, atol=2e-6, what='sum dA')
    # max (skip rows whose arg is ambiguous only if torch and the reference disagree: they do not here)
    _, E = oracle.spmm('max', rp, col, val, X)
    assert_close(oracle.spmm_mask(colptr, row, tval, G, E), g['max_dX'], rtol=1e-5, atol=2e-6, what='max dX')
    assert_close(oracle.sddmm_mask(rp, col, G, X, E), g['max_dA'], rtol=1e-5, atol=2e-6, what='max dA')
    if 'mean_dX' in g:
        deg = np.maximum(np.diff(rp), 1).astype(np.float32)
        sval = (val / np.repeat(deg, np.diff(rp)))[perm]
        dXm, _ = oracle.spmm('sum', colptr, row, sval, G)
        assert_close(dXm, g['mean_dX'], rtol=1e-5, atol=2e-6, what='mean dX')
        assert_close(oracle.sddmm(rp, col, G, X, reduce='mean'), g['mean_dA'], rtol=1e-5, atol=2e-6, what='mean dA')


def test_nan_and_identity_semantics():
    """gspmm.h:16-17 macros evaluated literally: MAX skips NaN; MIN keeps the *next* value after a NaN;
    all-below-identity rows return (float)INT_MIN with E=-1 (gspmm.h:133-146)."""
    rp = np.array([0, 3, 4], np.int32)
    col = np.array([0, 1, 2, 0], np.int32)
    val = np.ones(4, np.float32)
    X = np.array([[1.0], [np.nan], [5.0]], np.float32)
    C, E = oracle.spmm('max', rp, col, val, X)
    assert C[0, 0] == 5.0 and E[0, 0] == 2
    C, E = oracle.spmm('min', rp, col, val, X)
    assert C[0, 0] == 5.0 and E[0, 0] == 0  # literal macro behaviour
    X2 = np.full((3, 1), -3e9, np.float32)
    C, E = oracle.spmm('max', rp, col, val, X2)
    assert C[0, 0] == np.float32(-2147483648.0) and E[0, 0] == -1


def test_gspmm_oracle_vs_torch_emulation_fixture():
    """orc_gspmm_csr_f32 (restating src/gspmm-fp/gspmm.cu:212-245) against tests/golden/gspmm_dyadic_N8.npz, whose expected
    outputs come from plain torch (elementwise compute on gathered rows + Tensor.scatter_reduce_), see
    tests/golden/make_gspmm_golden.py.  Dyadic inputs make add/sub/mul exact in any order; div within 1e-5."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'gspmm_dyadic_N8.npz'))
    rp, col, val, X = z['rowptr'], z['col'], z['val'], z['X']
    assert (np.diff(rp) == 0).sum() >= 50
    for ri, red in enumerate(('sum', 'max', 'min', 'mean')):
        for ci, comp in enumerate(('add', 'sub', 'mul', 'div')):
            got = oracle.gspmm(ri, ci, rp, col, val, X)
            exp = z[f'{red}_{comp}']
            if comp == 'div' or red == 'mean':
                np.testing.assert_allclose(got, exp, rtol=1e-5, atol=1e-6, err_msg=f'{red} {comp}')
            else:
                assert np.array_equal(got, exp), f'{red} {comp}'  # exact values (+0 == -0: torch's amin/amax tie rule
                #                                                     for signed zeros is not the reference macro's)
