#!/usr/bin/env python3
"""What leaving the plan's column-slice order costs: the planned headline sum with rows longer than DGS_PLAN_NOCUT chunked
without column cuts (round 4: prices the hub rows that a strict chain takes out of the slice order).
    python bench/nocut_probe.py [feat]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dgsparse-lib_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

from bench import graphgen  # noqa: E402
from dgsparse import _capi  # noqa: E402


def t(fn, n=100):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = 1 << 20
rp, col, st = graphgen.powerlaw_csr(M, M * 16, alpha=2.1, dmax=1 << 16, cols='powerlaw', seed=0, device='cuda', as_torch=True)
g = torch.Generator(device='cuda')
g.manual_seed(1)
val = torch.rand(st['nnz'], generator=g, device='cuda')
X = torch.rand((st['K'], N), generator=g, device='cuda')
deg = (rp[1:] - rp[:-1]).long()
# part 1: hub chains OFF, rows above `nocut` leave the column-slice order but keep the tree: the pure locality price
os.environ['DGS_HUB_CHAIN'] = '0'
for nocut in (0, 32768, 16384, 8192, 4096, 2048, 1024):
    if nocut:
        os.environ['DGS_PLAN_NOCUT'] = str(nocut)
    else:
        os.environ.pop('DGS_PLAN_NOCUT', None)
    _capi.reload_tuning()
    plan = _capi.spmm_plan(rp, col, st['K'], N)
    ms = t(lambda: _capi.spmm(_capi.SUM, rp, col, val, X, plan=plan))
    m = deg > (nocut if nocut else 1 << 30)
    print(f'feat {N} nocut {nocut or "off"}: rows above {int(m.sum())}, nnz above {int(deg[m].sum())}: planned sum {ms:.4f} ms', flush=True)

# part 2: hub chains at several thresholds (plan built per threshold; plan-free call beside it): what the chains cost in all
os.environ.pop('DGS_PLAN_NOCUT', None)
for th in (0, 32768, 16384, 8192, 4096, 2048):
    os.environ['DGS_HUB_CHAIN'] = str(th)
    _capi.reload_tuning()
    plan = _capi.spmm_plan(rp, col, st['K'], N)
    ms = t(lambda: _capi.spmm(_capi.SUM, rp, col, val, X, plan=plan))
    ms0 = t(lambda: _capi.spmm(_capi.SUM, rp, col, val, X), n=50)
    m = deg > (th if th else 1 << 30)
    print(f'feat {N} hub chains above {th or "off"}: rows {int(m.sum())}, nnz {int(deg[m].sum())}, longest {int(deg.max())}: '
          f'planned sum {ms:.4f} ms, plan-free {ms0:.4f} ms', flush=True)
