#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ (run ONCE in the authoring container).

Expected outputs come from sources OUTSIDE this repo's oracle, so they pin it:
  * torch.sparse.mm(csr, X[, reduce]) on CPU -- the oracle the reference's own pytest asserts
    against (/root/reference/test/test_spmm.py:25,60,97,134) -- plus
    torch.ops.aten._sparse_mm_reduce_impl for the arg positions, from which the reference's E
    (arg COLUMN id, -1 for empty rows; include/cuda/spmm_cuda.cuh:38-41,52) is derived;
  * torch autograd through torch.sparse.mm for dX / dA (test_spmm.py:29-40);
  * the reference's own spmm_reference_host / sddmm_reference_host (example/util/sp_util.hpp:63-112)
    compiled in place into oracle/_ref (sequential fp32, no FMA);
  * scipy tocsc() for csr2csc (test/test_csr2csr.py:40-49), on synthetic graphs and on the CSR of the
    reference's fixture example/data/p2p-Gnutella31.mtx (data file; SNAP public dataset).
Nothing here is reference source text: inputs + expected outputs + seeds only.

Usage: python tests/golden/make_golden.py   (needs /root/reference and oracle/_ref built)
"""
import os
import sys

import numpy as np
import scipy.io
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (only oracle.ref_* = the reference's own loops are used here)
from bench import graphgen  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def torch_expected(rowptr, col, val, X, with_grads=False):
    """sum/mean/max/min outputs + E from torch CPU; optionally dX/dA grads for sum, mean, max."""
    M, K = rowptr.shape[0] - 1, X.shape[0]
    nnz = col.shape[0]
    out = {}
    crow = torch.from_numpy(rowptr.astype(np.int32))
    ccol = torch.from_numpy(col.astype(np.int32))
    for red, tred in (('sum', 'sum'), ('mean', 'mean'), ('max', 'amax'), ('min', 'amin')):
        v = torch.from_numpy(val.copy()).requires_grad_(with_grads)
        A = torch.sparse_csr_tensor(crow, ccol, v, size=(M, K))
        x = torch.from_numpy(X.copy()).requires_grad_(True)  # arg output needs requires_grad
        if red == 'sum':
            y = torch.sparse.mm(A, x)
        else:
            y = torch.sparse.mm(A, x, tred)
        out[f'{red}_out'] = y.detach().numpy().copy()
        if red in ('max', 'min'):
            _, arg = torch.ops.aten._sparse_mm_reduce_impl(A.detach(), x, tred)
            arg = arg.numpy().astype(np.int64)
            E = np.where(arg < nnz, col[np.minimum(arg, max(nnz - 1, 0))] if nnz else -1, -1).astype(np.int32)
            out[f'{red}_E'] = E
        if with_grads and red in ('sum', 'mean', 'max'):
            G = torch.from_numpy(graphgen.features(M, X.shape[1], seed=77))
            y.backward(G)
            out[f'{red}_dX'] = x.grad.numpy().copy()
            out[f'{red}_dA'] = v.grad.numpy().copy() if v.grad is not None else A.grad.values().numpy().copy()
            out['G'] = G.numpy()
    return out


def make_case(name, rowptr, col, val, X, grads=False, ref=True, sddmm=True, csc=True, x_seed=None,
              grad_reds=('sum', 'mean', 'max')):
    """x_seed: X == graphgen.features(K, N, x_seed) (+ optional shift recorded by the caller); big random
    inputs are then NOT stored - tests regenerate them (numpy PCG64 streams are version-stable) and
    check the stored CRC.  D1 = features(M, N, 55) and G = features(M, N, 77) are always regenerated."""
    import zlib
    M, K = rowptr.shape[0] - 1, X.shape[0]
    d = dict(rowptr=rowptr.astype(np.int32), col=col.astype(np.int32), val=val.astype(np.float32))
    if x_seed is None:
        d['X'] = X
    else:
        d['x_seed'] = np.int64(x_seed)
        d['x_shift'] = np.float32(X.flat[0] - graphgen.features(K, X.shape[1], x_seed).flat[0])
    d['x_crc'] = np.int64(zlib.crc32(np.ascontiguousarray(X).tobytes()))
    d['N'] = np.int64(X.shape[1])
    d['K'] = np.int64(K)
    ex = torch_expected(d['rowptr'], d['col'], d['val'], X, with_grads=grads)
    ex.pop('G', None)
    for k in list(ex):
        if k.endswith('_dX') or k.endswith('_dA'):
            if k.split('_')[0] not in grad_reds:
                ex.pop(k)
    d.update(ex)
    if ref:
        d['ref_sum_out'] = oracle.ref_spmm_sum(d['rowptr'], d['col'], d['val'], X)
    if sddmm:
        D1 = graphgen.features(M, X.shape[1], seed=55)
        d['ref_sddmm_out'] = oracle.ref_sddmm(d['rowptr'], d['col'], D1, X)
    if csc:
        A = sp.csr_matrix((d['val'], d['col'], d['rowptr']), shape=(M, K))
        # scipy tocsc is a stable counting sort; perm recovered by transposing the positions
        P = sp.csr_matrix((np.arange(1, col.shape[0] + 1, dtype=np.float64), d['col'], d['rowptr']), shape=(M, K)).tocsc()
        T = A.tocsc()
        if not np.array_equal(P.indptr, T.indptr):
            raise RuntimeError('scipy pattern mismatch')
        d['csc_colptr'] = T.indptr.astype(np.int32)
        d['csc_row'] = T.indices.astype(np.int32)
        d['csc_val'] = T.data.astype(np.float32)
        d['csc_perm'] = (P.data - 1).astype(np.int32)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **d)
    print(name, 'M', M, 'nnz', col.shape[0], 'N', X.shape[1], '->', os.path.getsize(os.path.join(OUT, name + '.npz')) // 1024, 'KiB')


def tiny_cases():
    """Hand-shaped: empty rows, 1-nnz rows, duplicate columns, ties, negative & zero products."""
    rowptr = np.array([0, 0, 1, 4, 4, 9, 12, 12], np.int32)  # rows 0,3,6 empty
    col = np.array([2, 0, 3, 3, 1, 1, 4, 5, 5, 6, 0, 2], np.int32)  # dup cols in rows 2 and 4
    val = np.array([.5, .1, .2, .2, 0., .1, .2, .2, .1, -1., 2., -.5], np.float32)
    for N in (1, 3, 32, 33, 64, 128):
        rng = np.random.Generator(np.random.PCG64(N))
        X = (rng.integers(0, 3, (7, N)) / 10).astype(np.float32)  # {0,.1,.2}: many exact ties
        X[5] = -X[5] - .1  # negative products
        make_case(f'tiny_N{N}', rowptr, col, val, X)
    # torch's CSR autograd rejects duplicate entries, so the gradient cases use a duplicate-free pattern
    col2 = np.array([2, 0, 3, 5, 1, 2, 4, 5, 6, 0, 2, 6], np.int32)  # sorted: torch autograd needs it
    for N in (3, 32):
        rng = np.random.Generator(np.random.PCG64(100 + N))
        X = (rng.integers(0, 3, (7, N)) / 10).astype(np.float32)
        X[5] = -X[5] - .1
        make_case(f'tiny_nodup_grad_N{N}', rowptr, col2, val, X, grads=True)


def main():
    assert oracle.have_ref(), 'build oracle/_ref first (make -C oracle ref)'
    tiny_cases()
    # Cora-shaped plumbing config (BASELINE.json configs[0]): unit weights, U[0,1) features, N=32
    rp, col, st = graphgen.dataset_shaped('cora', seed=0)
    make_case('cora_shaped_N32', rp, col, graphgen.weights(col.shape[0], 'ones'), graphgen.features(st['K'], 32, 0),
              grads=True, x_seed=0)
    # a 1024-row graph with random weights, N=64 (weights exercise the single-rounding product)
    rp, col, st = graphgen.powerlaw_csr(1024, 6000, alpha=2.5, dmax=200, seed=11)
    make_case('small_weighted_N64', rp, col, graphgen.weights(col.shape[0], 'uniform', 1), graphgen.features(st['K'], 64, 1),
              grads=True, x_seed=1, grad_reds=('sum', 'max'))
    # 4096-row power-law sample with one >10^4-nnz row, tied values, N=8
    rp, col, st = graphgen.powerlaw_csr(4096, 60000, alpha=2.0, dmax=4096, seed=3)
    deg = np.diff(rp)
    big = np.arange(0, 4096 * 3, 1, dtype=np.int64) % 4096  # a 12288-nnz row WITH duplicate columns
    r = 1234
    col = np.concatenate([col[:rp[r]], np.sort(big).astype(np.int32), col[rp[r + 1]:]])
    deg[r] = big.shape[0]
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    make_case('powerlaw4k_tied_N8', rp, col, graphgen.weights(col.shape[0], 'tied', 3),
              (np.random.Generator(np.random.PCG64(9)).integers(0, 3, (4096, 8)) / 10).astype(np.float32))
    # duplicate-free variant (full 4096-nnz row) for the gradient vectors
    rp2, col2, _ = graphgen.powerlaw_csr(4096, 60000, alpha=2.0, dmax=4096, seed=5)
    deg2 = np.diff(rp2)
    col2 = np.concatenate([col2[:rp2[r]], np.arange(4096, dtype=np.int32), col2[rp2[r + 1]:]])
    deg2[r] = 4096
    rp2 = np.concatenate([[0], np.cumsum(deg2)]).astype(np.int32)
    make_case('powerlaw4k_signed_grad_N8', rp2, col2, graphgen.weights(col2.shape[0], 'signed', 6),
              graphgen.features(4096, 8, 6) - np.float32(0.5), grads=True, x_seed=6, grad_reds=('sum', 'max'))
    # csr2csc on the reference's own fixture (test/test_csr2csr.py uses this file with scipy)
    mtx = '/root/reference/example/data/p2p-Gnutella31.mtx'
    A = scipy.io.mmread(mtx).astype('float32').tocsr()
    T = A.tocsc()
    np.savez_compressed(os.path.join(OUT, 'p2p_gnutella31_csr2csc.npz'), rowptr=A.indptr.astype(np.int32),
                        col=A.indices.astype(np.int32), val=A.data.astype(np.float32), shape=np.array(A.shape),
                        csc_colptr=T.indptr.astype(np.int32), csc_row=T.indices.astype(np.int32),
                        csc_val=T.data.astype(np.float32))
    print('p2p', os.path.getsize(os.path.join(OUT, 'p2p_gnutella31_csr2csc.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
