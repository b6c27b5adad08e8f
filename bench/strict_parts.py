#!/usr/bin/env python3
"""Where the strict-order schedule spends its time: the headline graph cut into row-length classes (each class as its
own matrix, other rows emptied), default vs strict.    python bench/strict_parts.py [feat]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dgsparse-lib_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

from bench import graphgen  # noqa: E402
from dgsparse import _capi  # noqa: E402


def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rp, col, st = graphgen.dataset_shaped('synth1m', seed=0, device='cuda', as_torch=True)
val = torch.rand(st['nnz'], device='cuda')
X = torch.rand((st['K'], N), device='cuda')
deg = (rp[1:] - rp[:-1]).long()
rowid = torch.repeat_interleave(torch.arange(st['M'], device='cuda'), deg)
for lo, hi in ((0, 64), (64, 256), (256, 2048), (2048, 8192), (8192, 1 << 30), (0, 1 << 30)):
    keep_row = (deg > lo) & (deg <= hi)
    keep = keep_row[rowid]
    d2 = torch.where(keep_row, deg, torch.zeros_like(deg))
    rp2 = torch.zeros(st['M'] + 1, dtype=torch.int64, device='cuda')
    rp2[1:] = torch.cumsum(d2, 0)
    rp2 = rp2.int()
    col2, val2 = col[keep].contiguous(), val[keep].contiguous()
    a = t(lambda: _capi.spmm(_capi.SUM, rp2, col2, val2, X))
    b = t(lambda: _capi.spmm(_capi.SUM, rp2, col2, val2, X, algorithm=_capi.ALG_STRICT_SUM))
    print(f'rows with {lo} < nnz <= {hi}: {int(keep_row.sum())} rows, {col2.numel()} nnz, longest {int(d2.max())}: default {a:.4f} ms, strict {b:.4f} ms')
