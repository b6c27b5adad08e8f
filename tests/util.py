"""Shared helpers for the test-suite: golden-case loader and comparison helpers."""
import glob
import os
import zlib

import numpy as np

from bench import graphgen

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden_names(pattern='*'):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, pattern + '.npz'))
                  if 'p2p' not in p)


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
