#!/usr/bin/env python3
"""The halo product of the bench workload's rank 3 of 8 (2^20 rows per GPU, ~16 nnz/row, locality 0.8, feat 64) on one
GPU: compact A_rem . B_halo added into C (a) as round 2 first did it - product into a temporary + scatter-add of its rows -
and (b) with the accumulating SpMM (dgs_spmm_csr_acc_f32).  python bench/acc_vs_scatter.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dgsparse-lib_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

from dgsparse import _capi, dist as dd  # noqa: E402

part = dd.synthetic_partition(3, 8, 1 << 20, 16, cols='powerlaw', locality=0.8, seed=0, device='cuda')
N = 64
eng = dd.DistSpMM(part, N, standalone=True)
plan = eng.plan
B_halo = torch.rand((eng.n_halo, N), device='cuda')
C = torch.rand((part.n_local, N), device='cuda')
rp, col, val = plan.rem
P = _capi.spmm_plan(rp, col, eng.n_halo, N)


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def a():
    Cr, _ = _capi.spmm(0, rp, col, val, B_halo, plan=P)
    _capi.scatter_add_rows(C, plan.rem_rows, Cr)


def b():
    _capi.spmm_acc(rp, col, val, B_halo, C, plan.rem_rows, plan=P)


print(f'rows with a remote entry {plan.rem_rows.numel()} of {part.n_local}, remote nnz {col.numel()}, halo rows {eng.n_halo}')
print(f'(a) product into a temporary + scatter-add: {timeit(a):.4f} ms')
print(f'(b) accumulating SpMM:                      {timeit(b):.4f} ms')

# max: the undivided product over [local | halo] columns (what the non-overlapped path runs after the exchange) against the
# two products of the overlapped path (local columns while the halo travels, then the accumulating max of the halo part)
B_ext = torch.rand((part.n_local + eng.n_halo, N), device='cuda')
erp, ecol, eval_ = part.rowptr, plan.col_ext, part.val
lrp, lcol, lval = plan.loc
Pe = _capi.spmm_plan(erp, ecol, B_ext.shape[0], N)
Pl = _capi.spmm_plan(lrp, lcol, part.n_local, N)
B_l, B_h = B_ext[:part.n_local], B_ext[part.n_local:]
Cm, Em = _capi.spmm(1, lrp, lcol, lval, B_l, plan=Pl)


def m_whole():
    _capi.spmm(1, erp, ecol, eval_, B_ext, plan=Pe)


def m_local():
    _capi.spmm(1, lrp, lcol, lval, B_l, plan=Pl)


def m_acc():
    _capi.spmm_acc_max(rp, col, val, B_h, Cm, Em, plan.rem_rows, part.n_local, part.n_local, plan.h_lo, plan=P)


tw, tl, ta = timeit(m_whole), timeit(m_local), timeit(m_acc)
print(f'max, undivided extended product:            {tw:.4f} ms')
print(f'max, local columns only:                    {tl:.4f} ms   (runs under the exchange)')
print(f'max, accumulating halo product:             {ta:.4f} ms   (what is left after the exchange; sum of both {tl + ta:.4f})')
