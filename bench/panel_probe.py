#!/usr/bin/env python3
"""Column-panel schedule (csrc/spmm_panel.h) vs the row-stream schedule on a dense, dataset-shaped graph.

    python bench/panel_probe.py [--graph reddit] [--feat 128] [--scale 1.0] [--kb 1024 1536 2048] [--lead 0 1 2]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dgsparse-lib_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

from bench import graphgen  # noqa: E402
from dgsparse import _capi  # noqa: E402


def timeit(fn, warm=3, iters=10):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--graph', default='reddit')
    ap.add_argument('--feat', type=int, default=128)
    ap.add_argument('--scale', type=float, default=1.0)
    ap.add_argument('--cols', default='powerlaw')
    ap.add_argument('--op', default='sum')
    ap.add_argument('--dmax', type=int, default=0, help='override the max degree of the shape')
    ap.add_argument('--deg', type=float, default=0, help='override the mean degree of the shape')
    ap.add_argument('--ncols', type=int, default=0, help='override K (columns of A = rows of the dense operand)')
    ap.add_argument('--kb', type=int, nargs='*', default=[1536])
    ap.add_argument('--lead', type=int, nargs='*', default=[1])
    a = ap.parse_args()
    op = {'sum': _capi.SUM, 'mean': _capi.MEAN, 'max': _capi.MAX, 'min': _capi.MIN, 'sddmm': -1}[a.op]
    if a.dmax or a.ncols or a.deg:
        sh = graphgen.SHAPES[a.graph]
        rowptr, col, st = graphgen.powerlaw_csr(int(sh['M'] * a.scale),
                                                int(sh['M'] * a.scale * a.deg) if a.deg else int(sh['nnz'] * a.scale),
                                                alpha=sh['alpha'],
                                                dmax=a.dmax or sh['dmax'], K=a.ncols or None, cols=a.cols, device='cuda', as_torch=True)
    else:
        rowptr, col, st = graphgen.dataset_shaped(a.graph, cols=a.cols, scale=a.scale, device='cuda', as_torch=True)
    M, nnz = st['M'], st['nnz']
    g = torch.Generator(device='cuda')
    g.manual_seed(1)
    B = torch.rand((st['K'], a.feat), device='cuda', generator=g)
    val = torch.rand(nnz, device='cuda', generator=g) + 0.5
    print(f'{a.graph}: M={M} nnz={nnz} max_deg={st["max_deg"]} feat={a.feat} op={a.op}', flush=True)
    if a.op == 'sddmm':
        D1 = torch.rand((M, a.feat), device='cuda', generator=g)

        def call():
            return _capi.sddmm(rowptr, col, D1, B), None
    else:
        def call():
            return _capi.spmm(op, rowptr, col, val, B)
    os.environ['DGS_PANEL'] = '0'
    _capi.reload_tuning()
    ref, _ = call()
    t0 = timeit(lambda: call())
    print(f'row-stream schedule : {t0:8.3f} ms  {2e-6 * nnz * a.feat / t0:8.1f} GFLOP/s', flush=True)
    os.environ['DGS_PANEL'] = '1'
    _capi.reload_tuning()
    for kb in a.kb:
        for lead in a.lead:
            os.environ['DGS_PANEL_KB'] = str(kb)
            _capi.reload_tuning()
            os.environ['DGS_PANEL_LEAD'] = str(lead)
            _capi.reload_tuning()
            out, _ = call()
            err = ((out - ref).abs() / (ref.abs() + 1e-3)).max().item()
            t = timeit(lambda: call())
            print(f'panel kb={kb:5d} lead={lead}: {t:8.3f} ms  {2e-6 * nnz * a.feat / t:8.1f} GFLOP/s  '
                  f'max rel diff vs row-stream {err:.2e}', flush=True)


if __name__ == '__main__':
    main()
