#!/usr/bin/env python3
"""How far the reference's own sequential fp32 chain (include/cuda/spmm_cuda.cuh:27-47; host twin example/util/sp_util.hpp:73-83)
is from the exact sum, by row length and VALUE LAW (round 5, CPU only; the oracle's chains vs a float64 sum):

  tied              weights and features in {0, .1, .2} - the reference's fill_random (example/util/sp_util.hpp:44-48): only
                    four distinct non-zero products
  tied_w_uniform_x  weights in {0, .1, .2}, features U[0,1)
  uniform           both U[0,1) (test/test_spmm.py:20)

With a handful of distinct addends the rounding error of `sum += t` is not a random walk: the same addend rounds the same way
for as long as the running sum stays in one binade, so the chain picks up a SYSTEMATIC bias - beyond 1e-5 of the exact sum from
~2 000 nnz on, whatever the order of any more accurate summation.     python experiments/chain_error_tied.py > profiles/r05_chain_error_by_value_law.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402

rng = np.random.default_rng(7)
K, N, R = 1 << 16, 64, 64
print(f'{R} rows per length, {N} features, columns uniform in {K}; relative error of the sequential fp32 chain vs the float64 sum')
for law in ('tied', 'tied_w_uniform_x', 'uniform'):
    print(law)
    for L in (256, 512, 1024, 1536, 2048, 3072, 4096, 5120, 6144, 7168, 8192, 10240, 12288, 16384, 32768):
        rp = (np.arange(R + 1) * L).astype(np.int32)
        col = np.sort(rng.integers(0, K, (R, L)), axis=1).astype(np.int32).ravel()
        tied_w = law != 'uniform'
        val = (rng.integers(0, 3, R * L) / 10).astype(np.float32) if tied_w else rng.random(R * L, dtype=np.float32)
        X = (rng.integers(0, 3, (K, N)) / 10).astype(np.float32) if law == 'tied' else rng.random((K, N), dtype=np.float32)
        seq = oracle.spmm('sum', rp, col, val, X, fma=False, threads=oracle.max_threads())[0]
        fma = oracle.spmm('sum', rp, col, val, X, fma=True, threads=oracle.max_threads())[0]
        ex = oracle.spmm_sum_f64(rp, col, val, X)
        e1 = np.abs(seq - ex) / np.maximum(np.abs(ex), 1e-6)
        e2 = np.abs(fma - ex) / np.maximum(np.abs(ex), 1e-6)
        print(f'  {L:6d} nnz   chain (mul, add): mean {e1.mean():.2e} max {e1.max():.2e}   fmaf chain: mean {e2.mean():.2e} max {e2.max():.2e}'
              f'   elements beyond 1e-5: {int((e1 > 1e-5).sum())} / {e1.size}')
