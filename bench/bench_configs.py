#!/usr/bin/env python3
"""Times every BASELINE.json configuration that fits one GPU (synthetic, dataset-shaped) and prints one JSON line
per config: GFLOP/s (2*nnz*N), algorithmic GB/s and fraction of the 8 TB/s HBM roofline.
Protocol: 10 warm-up + 100 timed launches between HIP events, median of 3 repeats.

    python bench/bench_configs.py [--quick] [--only NAME]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dgsparse-lib_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

from bench import graphgen  # noqa: E402
from dgsparse import _capi  # noqa: E402

PEAK = 8000.0


def timeit(fn, warm=10, iters=100, reps=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e-3)
    return sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--quick', action='store_true')
    ap.add_argument('--only', default='')
    ap.add_argument('--no-plan', action='store_true', help='plan-free calls only (default: cached locality plan where one applies)')
    ap.add_argument('--cols', default='powerlaw', choices=['powerlaw', 'uniform', 'local'],
                    help="column structure of the synthetic graphs: 'local' = community-like (columns in a window around the row id, a "
                         "graph in a locality-preserving order), what the XCD-aware mappings and the plan's column slices are for")
    a = ap.parse_args()
    it = 20 if a.quick else 100
    cfgs = [  # (label, graph, op, feat)
        ('C2 arxiv-shaped SpMM-sum', 'arxiv', 'sum', 64),
        ('C3 reddit-shaped SpMM-sum', 'reddit', 'sum', 128),
        ('C3 reddit-shaped SpMM-max', 'reddit', 'max', 128),
        ('C3 reddit-shaped SDDMM', 'reddit', 'sddmm', 64),
        ('C4 products-shaped SDDMM', 'products', 'sddmm', 64),
        ('NS synth-1M SpMM-sum', 'synth1m', 'sum', 32),
        ('NS synth-1M SpMM-sum', 'synth1m', 'sum', 64),
        ('NS synth-1M SpMM-sum', 'synth1m', 'sum', 128),
        ('NS synth-1M SpMM-max', 'synth1m', 'max', 64),
        ('NS synth-1M SpMM-mean', 'synth1m', 'mean', 64),
        ('NS synth-1M SDDMM', 'synth1m', 'sddmm', 64),
        ('small cora-shaped SpMM-sum', 'cora', 'sum', 64),
        ('small pubmed-shaped SpMM-sum', 'pubmed', 'sum', 64),
    ]
    cache = {}
    for label, gname, op, N in cfgs:
        if a.only and a.only not in label and a.only != gname:
            continue
        if gname not in cache:
            cache.clear()
            torch.cuda.empty_cache()
            cache[gname] = graphgen.dataset_shaped(gname, seed=0, cols=a.cols, device='cuda', as_torch=True)
        rp, col, st = cache[gname]
        M, K, nnz = st['M'], st['K'], st['nnz']
        g = torch.Generator(device='cuda')
        g.manual_seed(1)
        val = torch.rand(nnz, generator=g, device='cuda')
        X = torch.rand((K, N), generator=g, device='cuda')
        sched = ''
        if op == 'sddmm':
            D1 = torch.rand((M, N), generator=g, device='cuda')
            plan = None if a.no_plan else _capi.spmm_plan(rp, col, K, N)
            fused = plan is not None and _capi._lib.dgs_sddmm_csr_schedule(M, K, N, nnz, 0) != 2 and \
                plan.info.n_pslots * 256 >= nnz  # the library's own dispatch rule (csrc/sddmm.hip)
            sched = ('panel' if _capi._lib.dgs_sddmm_csr_schedule(M, K, N, nnz, 0) == 2 else
                     'fused rows+units over the plan' if fused else 'nnz-balanced')
            t = timeit(lambda: _capi.sddmm(rp, col, D1, X, plan=plan), iters=it)
            balg = 4 * (M + 1) + 8 * nnz + 4 * (M + K) * N
        else:
            o = {'sum': 0, 'max': 1, 'min': 2, 'mean': 3}[op]
            plan = None if a.no_plan else _capi.spmm_plan(rp, col, K, N)
            sched = _capi.spmm_schedule(o, M, K, N, nnz) + ('+plan' if plan is not None else '')
            t = timeit(lambda: _capi.spmm(o, rp, col, val, X, plan=plan), iters=it)
            balg = 4 * (M + 1) + 8 * nnz + 4 * K * N + 4 * M * N * (2 if op in ('max', 'min') else 1)
        print(json.dumps(dict(config=label, graph=gname, cols=a.cols, M=M, nnz=nnz, max_deg=st['max_deg'], op=op, feat=N, schedule=sched,
                              us=round(t * 1e6, 2), gflops=round(2.0 * nnz * N / t / 1e9, 1),
                              alg_gbs=round(balg / t / 1e9, 1), frac=round(balg / t / 1e9 / PEAK, 4))), flush=True)


if __name__ == '__main__':
    main()
