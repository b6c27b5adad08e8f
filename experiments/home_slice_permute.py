#!/usr/bin/env python3
"""Upper bound of the 'home-slice order for short rows' lever (DESIGN 4.1d / VERDICT r2 #5a) without touching a kernel: the
rows of the headline matrix are PHYSICALLY permuted so that the contiguous eighth of the row blocks each XCD walks holds the
rows whose cold columns fall mostly into ONE column slice, sorted by their smallest cold column of that slice
(experiments/l2_model/home_slice_rows.py: 43 % -> 49.6 % hits in the LRU model).  Same nnz, same kernel, planned call."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dgsparse-lib_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

from bench import graphgen  # noqa: E402
from dgsparse import _capi  # noqa: E402


def t(fn, n=100):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def permute_rows(rp, col, val, order):
    deg = (rp[1:] - rp[:-1]).long()
    nd = deg[order]
    nrp = torch.zeros(rp.numel(), dtype=torch.int64, device=rp.device)
    nrp[1:] = torch.cumsum(nd, 0)
    src0 = rp[:-1].long()[order]
    idx = torch.repeat_interleave(src0 - nrp[:-1], nd) + torch.arange(int(nrp[-1]), device=rp.device)
    return nrp.int(), col[idx].contiguous(), val[idx].contiguous()


N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rp, col, st = graphgen.dataset_shaped('synth1m', seed=0, device='cuda', as_torch=True)
M, K, nnz = st['M'], st['K'], st['nnz']
val = torch.rand(nnz, device='cuda')
X = torch.rand((K, N), device='cuda')
plan = _capi.spmm_plan(rp, col, K, N)
base = t(lambda: _capi.spmm(_capi.SUM, rp, col, val, X, plan=plan))
print(f'original order            : {base * 1e3:.1f} us')

cnt = torch.bincount(col.long(), minlength=K)
cum = torch.cumsum(cnt, 0)
bounds = torch.searchsorted(cum, torch.tensor([nnz * x // 8 for x in range(1, 8)], device='cuda'))
rank = torch.empty(K, dtype=torch.long, device='cuda')
rank[torch.argsort(-cnt, stable=True)] = torch.arange(K, device='cuda')
cold = rank[col.long()] >= 16384
sl = torch.bucketize(col.long(), bounds, right=True)  # slice of every nnz
deg = (rp[1:] - rp[:-1]).long()
row_of = torch.repeat_interleave(torch.arange(M, device='cuda'), deg)
cnts = torch.zeros((M, 8), device='cuda')
cnts.index_put_((row_of[cold], sl[cold]), torch.ones(int(cold.sum()), device='cuda'), accumulate=True)
home = (cnts + torch.rand((M, 8), device='cuda') * 0.5).argmax(1)
key = torch.full((M,), K, dtype=torch.long, device='cuda')
ch = cold & (sl == home[row_of])
key.scatter_reduce_(0, row_of[ch], col.long()[ch], reduce='amin')
for name, order in (
        ('random row order         ', torch.randperm(M, device='cuda')),
        ('home slice, natural order', torch.argsort(home * M + torch.arange(M, device='cuda'))),
        ('home slice, sorted by key', torch.argsort(home * (K + 1) * 1 + key.double() / (K + 1), stable=True) if False else
         torch.argsort(home.double() * (K + 2) + key.double(), stable=True))):
    rp2, col2, val2 = permute_rows(rp, col, val, order)
    plan2 = _capi.spmm_plan(rp2, col2, K, N)
    ms = t(lambda: _capi.spmm(_capi.SUM, rp2, col2, val2, X, plan=plan2))
    print(f'{name}: {ms * 1e3:.1f} us   (rows per home slice: {torch.bincount(home, minlength=8).tolist() if "home" in name else "-"})')
