"""Shared helpers for the test-suite: golden-case loader and comparison helpers."""
import glob
import os
import zlib

import numpy as np

from bench import graphgen

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden_names(pattern='*'):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, pattern + '.npz'))
                  if 'p2p' not in p and 'gspmm' not in p and 'ca_condmat' not in p)


def load_golden(name):
    """Returns a dict; X / D1 / G are regenerated from their seeds when not stored (CRC-checked)."""
    z = dict(np.load(os.path.join(GOLDEN, name + '.npz')))
    if 'N' not in z:
        return z
    N, K = int(z['N']), int(z['K'])
    M = z['rowptr'].shape[0] - 1
    if 'X' not in z:
        z['X'] = (graphgen.features(K, N, int(z['x_seed'])) + z['x_shift']).astype(np.float32)
    assert zlib.crc32(np.ascontiguousarray(z['X']).tobytes()) == int(z['x_crc']), 'regenerated X drifted'
    z['D1'] = graphgen.features(M, N, seed=55)
    z['G'] = graphgen.features(M, N, seed=77)
    return z


def assert_close(a, b, rtol=1e-5, atol=1e-6, what=''):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bad = ~np.isclose(a, b, rtol=rtol, atol=atol, equal_nan=True)
    if bad.any():
        i = np.argwhere(bad)[0]
        raise AssertionError(f'{what}: {bad.sum()} / {bad.size} mismatches; first at {tuple(i)}: '
                             f'{a[tuple(i)]!r} vs {b[tuple(i)]!r}')


def assert_bitexact(a, b, what=''):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
    if a.dtype.kind == 'f':
        ai, bi = a.view(np.int32), b.view(np.int32)
    else:
        ai, bi = a, b
    bad = ai != bi
    if bad.any():
        i = np.argwhere(bad)[0]
        raise AssertionError(f'{what}: {bad.sum()} / {bad.size} bit mismatches; first at {tuple(i)}: '
                             f'{a[tuple(i)]!r} vs {b[tuple(i)]!r}')


def assert_sum_parity(C, Cseq, C64, S64=None, rtol=1e-5, atol=2e-6, what='', lens=None):
    """north_star bar for sum/mean: within 1e-5 relative of the reference's sequential fp32 result.  Where that
    cannot be meaningful -- rows with thousands of nnz, whose sequential fp32 chain is itself further than 1e-5
    from the exact (float64) sum, or cancelling signed data -- the result must instead be at least as close to the
    exact value as the sequential chain is, or within gamma of the row's condition scale S = sum|w*x|, with
    gamma = max(1e-6, eps/2 * sqrt(n)) for a row of n terms (the statistical error of ANY fp32 summation order; for
    the reference's non-negative test data S == |C| and n <= 1e4, i.e. at least three times tighter than 1e-5)."""
    C = np.asarray(C, np.float64)
    Cseq = np.asarray(Cseq, np.float64)
    ok = np.isclose(C, Cseq, rtol=rtol, atol=atol)
    err_gpu = np.abs(C - C64)
    err_seq = np.abs(Cseq - C64)
    scale = np.abs(C64) if S64 is None else S64
    gamma = 1e-6
    if lens is not None:
        gamma = np.maximum(1e-6, 3e-8 * np.sqrt(np.asarray(lens, np.float64)))[:, None]
    ok |= err_gpu <= np.maximum(err_seq, gamma * scale)
    if not ok.all():
        i = tuple(np.argwhere(~ok)[0])
        raise AssertionError(f'{what}: {(~ok).sum()} / {ok.size} outside the sum bar; first at {i}: gpu {C[i]!r} '
                             f'seq {Cseq[i]!r} exact {C64[i]!r}')


def around_matrix(rp, col, val, deg, a, b, n_out, compact=True):
    """The "around" form of the halo matrix for local columns [a, b) (dgs_spmm_csr_acc_min_around_f32): per row with a halo
    entry [slots of the columns < a | virtual entry a + row, weight 1, if the row has a local entry | slots of the columns >= b,
    shifted by n_out virtual ids].  Returns (rowptr, col, val, rows it touches); compact=False keeps every row (rows without halo
    entries then hold their virtual entry alone, or nothing)."""
    M = rp.size - 1
    rows = np.repeat(np.arange(M), deg)
    is_loc = (col >= a) & (col < b)
    has_loc = np.bincount(rows[is_loc], minlength=M) > 0
    has_rem = np.bincount(rows[~is_loc], minlength=M) > 0
    keep = np.nonzero(has_rem)[0] if compact else np.arange(M)
    nl = b - a
    # the halo entries in the around id space, and the virtual entries of the kept rows that have local columns
    hr, hc, hv = rows[~is_loc], col[~is_loc], val[~is_loc]
    hid = np.where(hc < a, hc, hc - nl + n_out)  # slot = column without the local block; ids >= a + n_out follow the virtual block
    vr = keep[has_loc[keep]]
    ar = np.concatenate([hr, vr])
    ac = np.concatenate([hid, a + vr])
    av = np.concatenate([hv, np.ones(vr.size, np.float32)])
    order = np.lexsort((ac, ar))
    ar, ac, av = ar[order], ac[order], av[order]
    cnt = np.bincount(ar, minlength=M)[keep]
    rpp = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    return rpp, np.ascontiguousarray(ac.astype(np.int32)), np.ascontiguousarray(av.astype(np.float32)), keep.astype(np.int32)
