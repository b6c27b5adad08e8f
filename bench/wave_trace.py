#!/usr/bin/env python3
"""Per-wave timeline of the row blocks of spmm_fused (rowptr -> (col,val) tile -> piece boundaries -> gather loop ->
tail), from wall_clock64() stamps the kernel writes when the library is built with -DDGS_TRACE=1:

    make -C dgsparse-lib_amd/csrc ARCH=gfx950 DEFS=-DDGS_TRACE=1 && python bench/wave_trace.py [arxiv|synth1m ...]

(the default build has no stamps and no dgs_debug_trace symbol).  Prints wave lifetimes, the mean of each phase, when
waves start, and how many row waves are alive chip-wide over the launch."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dgsparse-lib_amd')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import graphgen  # noqa: E402
from dgsparse import _capi  # noqa: E402

lib = _capi._lib
if not hasattr(lib, 'dgs_debug_trace'):
    raise SystemExit('library built without -DDGS_TRACE=1')
lib.dgs_debug_trace.argtypes = [ctypes.c_void_p]
lib.dgs_debug_trace.restype = None
N = 64
for gname in (sys.argv[1:] or ['arxiv', 'synth1m']):
    rp, col, st = graphgen.dataset_shaped(gname, seed=0, device='cuda', as_torch=True)
    g = torch.Generator(device='cuda')
    g.manual_seed(1)
    val = torch.rand(st['nnz'], generator=g, device='cuda')
    X = torch.rand((st['K'], N), generator=g, device='cuda')
    plan = _capi.spmm_plan(rp, col, st['K'], N)
    for _ in range(5):
        _capi.spmm(0, rp, col, val, X, plan=plan)
    nw = 4 * 200000
    buf = torch.zeros(nw * 8, dtype=torch.int64, device='cuda')
    torch.cuda.synchronize()
    lib.dgs_debug_trace(ctypes.c_void_p(buf.data_ptr()))
    _capi.spmm(0, rp, col, val, X, plan=plan)
    torch.cuda.synchronize()
    lib.dgs_debug_trace(None)
    t = buf.cpu().numpy().reshape(nw, 8)
    ids = np.nonzero(t[:, 0] > 0)[0]
    tt = t[ids].astype(np.float64) / 100.0  # 100 MHz -> us
    t0 = tt[:, 0].min()
    start, end = tt[:, 0] - t0, tt[:, 5] - t0
    life = end - start
    print(f'{gname}: {len(ids)} row waves, lifetime mean {life.mean():.2f} us (p95 {np.percentile(life, 95):.2f}); starts p10 '
          f'{np.percentile(start, 10):.1f} p50 {np.percentile(start, 50):.1f} p90 {np.percentile(start, 90):.1f} us; last end {end.max():.1f} us')
    d = np.diff(tt[:, :6], axis=1).mean(0)
    print('   phase means (us): rowptr + row table %.2f | (col,val) tile %.2f | piece boundaries %.2f | gather loop %.2f | tail %.2f' % tuple(d))
    grid = np.linspace(0, end.max(), 21)
    print('   row waves alive chip-wide at 5 % steps of the launch:', [int((start <= q).sum() - (end <= q).sum()) for q in grid])
