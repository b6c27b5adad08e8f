#!/usr/bin/env python3
"""First contact for the hub-chain code (round 4): strict and default (hub rows chained) against the oracle's fmaf chain on a
small graph with a few hub rows, every feature mapping.  Exit code 1 on the first mismatch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dgsparse-lib_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle  # noqa: E402
from bench import graphgen  # noqa: E402
from dgsparse import _capi  # noqa: E402

dev = 'cuda'
bad = 0
for M, nnz, dmax, hub in ((70000, 900000, 30000, 2000), (300000, 3000000, 60000, 16384)):
    rp, col, st = graphgen.powerlaw_csr(M, nnz, alpha=2.0, dmax=dmax, seed=3)
    lens = np.diff(rp)
    print(f'graph {M} rows, {col.shape[0]} nnz, max row {lens.max()}, rows > {hub}: {(lens > hub).sum()}', flush=True)
    val = graphgen.weights(col.shape[0], 'uniform', 3)
    os.environ['DGS_HUB_CHAIN'] = str(hub)
    _capi.reload_tuning()
    drp, dcol, dval = (torch.as_tensor(x, device=dev) for x in (rp, col, val))
    for N in (64, 16, 32, 128, 256, 384):
        X = graphgen.features(st['K'], N, 4)
        dX = torch.as_tensor(X, device=dev)
        ref, _ = oracle.spmm('sum', rp, col, val, X, fma=True, threads=oracle.max_threads())
        hubrows = lens > max(hub, 1024)
        plan = _capi.spmm_plan(drp, dcol, st['K'], N)
        for name, kw in (('strict', dict(algorithm=_capi.ALG_STRICT_SUM)), ('default', {}), ('planned', dict(plan=plan))):
            if name == 'planned' and plan is None:
                continue
            C, _ = _capi.spmm(_capi.SUM, drp, dcol, dval, dX, **kw)
            torch.cuda.synchronize()
            C = C.cpu().numpy()
            rows = slice(None) if name == 'strict' else hubrows
            eq = np.array_equal(C[rows].view(np.int32), ref[rows].view(np.int32))
            rel = np.abs(C - ref) / np.maximum(np.abs(ref), 1e-6)
            print(f'  N={N:4d} {name:8s} hub rows bit-exact: {eq}   max rel err all rows {rel.max():.2e}'
                  + (f'  plan n_hub={plan.info.n_hub}' if name == 'planned' else ''), flush=True)
            if not eq or rel.max() > 1e-5:
                bad += 1
sys.exit(1 if bad else 0)
