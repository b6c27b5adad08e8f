#!/usr/bin/env python3
"""One rank's shard of BASELINE.json's 8-GPU configuration C5 (16M x 16M, ~32 nnz/row, feat 64) on ONE GPU: rank 3 of 8,
2^21 rows, ~2^26 nnz, relabelled into the [local | halo] operand (halo rows filled locally instead of exchanged), timed
through DistSpMM.compute (the product after the exchange) with and without the cached plan.  python bench/c5_shard.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dgsparse-lib_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

from dgsparse import dist as dd  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for loc in (0.8, 0.125):
    part = dd.synthetic_partition(3, 8, 1 << 21, 32, cols='powerlaw' if loc > 0.5 else 'uniform', locality=loc, seed=0,
                                  device='cuda')
    N = 64
    eng = dd.DistSpMM(part, N, standalone=True)
    eng.B_ext.copy_(torch.rand(eng.B_ext.shape, device='cuda'))
    M, nnz, K = part.n_local, part.nnz, eng.B_ext.shape[0]
    res = dict(rows=M, nnz=nnz, halo_rows=eng.n_halo, operand_rows=K, locality=loc, halo_bytes=eng.n_halo * N * 4)
    for plan in ('1', '0'):
        os.environ['DGS_PLAN'] = plan
        eng.ops._plans.clear()
        for red in ('sum', 'max'):
            ms = timeit(lambda: eng.compute(red))
            alg = 4 * (M + 1) + 8 * nnz + 4 * K * N + 4 * M * N * (2 if red == 'max' else 1)
            res[f'{red}_plan{plan}'] = dict(ms=round(ms, 3), gflops=round(2.0 * nnz * N / ms / 1e6, 1),
                                            frac=round(alg / (ms * 1e-3) / 8e12, 4))
    print(json.dumps(res), flush=True)
    del eng, part
    torch.cuda.empty_cache()
