#!/usr/bin/env python3
"""Counterpart of the reference's benchmark harness (benchmark/bench_spmm_time.py:1-464): wall-clock time of 100
forward (and forward+backward) iterations of dgsparse.spmm_{sum,max,min,mean} after 10 warm-ups, algorithm 0, on
dataset-shaped synthetic graphs (no downloads here), feat in {32, 64, 128}.  The reference compares against
torch_sparse and DGL (not installable here); the comparison library on ROCm is ``torch.sparse.mm`` (hipSPARSE),
which only implements reduce=sum on the GPU.

    python bench/bench_spmm_time.py [--datasets cora citeseer pubmed ppi0 reddit] [--feats 32 64 128] [--json out]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dgsparse-lib_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
import warnings  # noqa: E402

import torch  # noqa: E402

import dgsparse  # noqa: E402
from bench import graphgen  # noqa: E402

warnings.filterwarnings('ignore', message='Sparse CSR tensor support is in beta')
OPS = {'sum': dgsparse.spmm_sum, 'max': dgsparse.spmm_max, 'min': dgsparse.spmm_min, 'mean': dgsparse.spmm_mean}


ITERS = 100


def clock(fn, warm=10, iters=None):
    iters = ITERS if iters is None else iters
    warm = min(warm, iters)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return time.time() - t  # seconds for `iters` iterations, as the reference reports (bench_spmm_time.py:38-45)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--datasets', nargs='+', default=['cora', 'citeseer', 'pubmed', 'ppi0'])
    ap.add_argument('--feats', nargs='+', type=int, default=[32, 64, 128])
    ap.add_argument('--json', default='')
    ap.add_argument('--iters', type=int, default=100, help='timed iterations (the reference uses 100)')
    ap.add_argument('--no-baseline', action='store_true', help='skip the torch.sparse.mm (hipSPARSE) comparison')
    a = ap.parse_args()
    global ITERS
    ITERS = a.iters
    rows = []
    for name in a.datasets:
        rp, col, st = graphgen.dataset_shaped(name, seed=0, device='cuda', as_torch=True)
        val = torch.ones(st['nnz'], device='cuda')  # weights = ones (test/utils.py:52)
        tcsr = torch.sparse_csr_tensor(rp, col, val, size=(st['M'], st['K']))
        dcsr = dgsparse.SparseTensor.from_torch_sparse_csr_tensor(tcsr.clone().detach(), True, requires_grad=True)
        for N in a.feats:
            X = torch.rand((st['K'], N), device='cuda', requires_grad=True)
            for red, fn in OPS.items():
                with torch.no_grad():
                    fwd = clock(lambda: fn(dcsr, X, 0))

                def fb():
                    out = fn(dcsr, X, 0)
                    out.sum().backward()
                    X.grad = None
                    dcsr.storage._values.grad = None

                bwd = clock(fb)
                row = dict(dataset=name, nodes=st['M'], nnz=st['nnz'], feat=N, reduce=red,
                           iters=ITERS,
                           dgsparse_fwd_s_per_100=round(fwd * 100 / ITERS, 6),
                           dgsparse_fwdbwd_s_per_100=round(bwd * 100 / ITERS, 6))
                if red == 'sum' and not a.no_baseline:
                    Xd = X.detach()
                    row['torch_sparse_mm_fwd_s_per_100'] = round(clock(lambda: torch.sparse.mm(tcsr, Xd)) * 100 / ITERS, 6)
                rows.append(row)
                print(json.dumps(row), flush=True)
    if a.json:
        json.dump(rows, open(a.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
