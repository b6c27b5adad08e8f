#!/usr/bin/env python3
"""Time of dgs_spmm_plan_build (once per matrix) on the dataset-shaped graphs: python bench/plan_build_time.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dgsparse-lib_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

from bench import graphgen  # noqa: E402
from dgsparse import _capi  # noqa: E402

for name in ('arxiv', 'synth1m', 'products'):
    rp, col, st = graphgen.dataset_shaped(name, seed=0, device='cuda', as_torch=True)
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan = _capi.spmm_plan(rp, col, st['K'], 64, force=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{name}: nnz {st['nnz']}, plan build {best * 1e3:.2f} ms, plan buffer {plan.buf.numel() / 1e6:.1f} MB, {plan}")
