#!/usr/bin/env python3
"""Where do the ~50 us of the arxiv-shaped SpMM (C2) go?  Times the call on variants of the graph (hub rows capped,
uniform degrees) and under DGS_MIN_WAVES settings, next to a plain copy of the same algorithmic bytes.
    python bench/arxiv_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dgsparse-lib_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

from bench import graphgen  # noqa: E402
from bench.bench_configs import timeit  # noqa: E402
from dgsparse import _capi  # noqa: E402

N = 64
s = graphgen.SHAPES['arxiv']


def run(tag, rp, col, st, plan_on=True):
    M, K, nnz = st['M'], st['K'], st['nnz']
    g = torch.Generator(device='cuda')
    g.manual_seed(1)
    val = torch.rand(nnz, generator=g, device='cuda')
    X = torch.rand((K, N), generator=g, device='cuda')
    plan = _capi.spmm_plan(rp, col, K, N) if plan_on else None
    t = timeit(lambda: _capi.spmm(0, rp, col, val, X, plan=plan), iters=200)
    print(f'{tag:44s} M {M} nnz {nnz} maxdeg {st["max_deg"]} plan {plan is not None and plan_on}: {t * 1e6:7.2f} us', flush=True)


variants = {
    'arxiv-shaped': dict(alpha=s['alpha'], dmax=s['dmax']),
    'same, degrees capped at 256': dict(alpha=s['alpha'], dmax=256),
    'same, degrees capped at 64': dict(alpha=s['alpha'], dmax=64),
}
graphs = {}
for k, v in variants.items():
    graphs[k] = graphgen.powerlaw_csr(s['M'], s['nnz'], alpha=v['alpha'], dmax=v['dmax'], seed=0, device='cuda', as_torch=True)
for k, (rp, col, st) in graphs.items():
    run(k, rp, col, st, True)
    run(k + ' (plan-free)', rp, col, st, False)
rp, col, st = graphs['arxiv-shaped']
for mw in (1024, 2048, 4096, 8192, 16384, 32768):
    os.environ['DGS_MIN_WAVES'] = str(mw)
    _capi.reload_tuning()
    run(f'arxiv-shaped DGS_MIN_WAVES={mw}', rp, col, st, True)
os.environ.pop('DGS_MIN_WAVES')
_capi.reload_tuning()
# a copy with the same algorithmic bytes (97 MB: 48 read + 48 written), and an empty-ish launch
a = torch.rand(12 * 1024 * 1024, device='cuda')
b = torch.empty_like(a)
print(f'copy of 48 MB -> 48 MB: {timeit(lambda: b.copy_(a), iters=200) * 1e6:.2f} us')
z = torch.zeros(64, device='cuda')
print(f'tiny elementwise launch: {timeit(lambda: z.add_(1.0), iters=200) * 1e6:.2f} us')
