#!/usr/bin/env python3
"""Headline graph (and others): default / planned / strict-fma / strict-nofma step times through the C ABI.
    python bench/strict_time.py [graph] [feat]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dgsparse-lib_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

from bench import graphgen  # noqa: E402
from dgsparse import _capi  # noqa: E402


def t(fn, n=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


name = sys.argv[1] if len(sys.argv) > 1 else 'synth1m'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rp, col, st = graphgen.dataset_shaped(name, seed=0, device='cuda', as_torch=True)
g = torch.Generator(device='cuda')
g.manual_seed(1)
val = torch.rand(st['nnz'], generator=g, device='cuda')
X = torch.rand((st['K'], N), generator=g, device='cuda')
plan = _capi.spmm_plan(rp, col, st['K'], N)
print(name, st['M'], st['nnz'], 'max_deg', st['max_deg'], 'feat', N)
print('default plan-free  ms', round(t(lambda: _capi.spmm(_capi.SUM, rp, col, val, X)), 4))
if plan is not None:
    print('default plan       ms', round(t(lambda: _capi.spmm(_capi.SUM, rp, col, val, X, plan=plan)), 4))
print('strict fma         ms', round(t(lambda: _capi.spmm(_capi.SUM, rp, col, val, X, algorithm=_capi.ALG_STRICT_SUM)), 4))
print('strict nofma       ms', round(t(lambda: _capi.spmm(_capi.SUM, rp, col, val, X, algorithm=_capi.ALG_STRICT_NOFMA)), 4))
print('strict fma no-val  ms', round(t(lambda: _capi.spmm(_capi.SUM, rp, col, None, X, algorithm=_capi.ALG_STRICT_SUM)), 4))
