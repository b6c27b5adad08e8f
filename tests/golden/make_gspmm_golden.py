#!/usr/bin/env python3
"""Golden vectors for the gspmm-fp surface (GSpMM_u_e: reduce in {sum,max,min,mean} x compute in {add,sub,mul,div}),
so that oracle/dgs_oracle.c:orc_gspmm_csr_f32 is pinned by something OUTSIDE this repo: the expected outputs come from
plain torch on the CPU -- the per-edge compute as elementwise tensor ops on gathered rows, the per-row reduce through
``Tensor.scatter_reduce_`` (torch's own segment reduction) -- following the semantics of the reference's
weightedSimpleSPMMKernel (src/gspmm-fp/gspmm.cu:212-245: sub = feature - value, div = feature / value, empty rows 0,
mean divides by the row length).  Values are small dyadic rationals: every sum is exact in fp32, so the fixture is
independent of the summation order.  Usage: python tests/golden/make_gspmm_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import graphgen  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    M, K, N = 400, 300, 8
    rp, col, st = graphgen.powerlaw_csr(M, 5000, K=K, alpha=2.0, dmax=150, seed=42, dedup=False)
    rp = np.concatenate([rp[:200], np.full(50, rp[200], rp.dtype), rp[200:]])  # 50 empty rows in the middle
    M = rp.shape[0] - 1
    rng = np.random.default_rng(42)
    val = (rng.integers(1, 9, col.shape[0]) / 4).astype(np.float32)           # {0.25 .. 2.0}: never 0 (div)
    val *= rng.choice([-1.0, 1.0], col.shape[0]).astype(np.float32)
    X = (rng.integers(-16, 17, (K, N)) / 8).astype(np.float32)
    rows = torch.from_numpy(np.repeat(np.arange(M), np.diff(rp)))
    a = torch.from_numpy(val)[:, None]
    b = torch.from_numpy(X)[torch.from_numpy(col).long()]
    lens = torch.from_numpy(np.diff(rp)).float()
    out = dict(rowptr=rp, col=col, val=val, X=X, N=N, K=K)
    for cname, t in (('add', a + b), ('sub', b - a), ('mul', a * b), ('div', b / a)):
        idx = rows[:, None].expand(-1, N)
        for rname, red in (('sum', 'sum'), ('max', 'amax'), ('min', 'amin'), ('mean', 'sum')):
            y = torch.zeros(M, N)
            y.scatter_reduce_(0, idx, t, red, include_self=False)   # rows without entries keep 0
            if rname == 'mean':
                y = y / lens.clamp(min=1)[:, None]
            out[f'{rname}_{cname}'] = y.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, 'gspmm_dyadic_N8.npz'), **out)
    print('wrote gspmm_dyadic_N8.npz', st)


if __name__ == '__main__':
    main()
