#!/usr/bin/env python3
"""What a plan costs a caller: first public-operator call on a FRESH SparseTensor vs the plan-free C-ABI
call, and a 20-call loop on a fresh tensor (plan built asynchronously from the 4th use on) vs the same loop plan-free and
with the plan already there.    python bench/plan_lifecycle.py [graph] [feat]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dgsparse-lib_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import dgsparse  # noqa: E402
from bench import graphgen  # noqa: E402
from dgsparse import _capi  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'synth1m'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rp, col, st = graphgen.dataset_shaped(name, seed=0, device='cuda', as_torch=True)
val = torch.rand(st['nnz'], device='cuda')
X = torch.rand((st['K'], N), device='cuda')


def wall(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def fresh():
    return dgsparse.SparseTensor(rowptr=rp.clone(), col=col.clone(), values=val, has_value=True)


with torch.no_grad():
    for _ in range(20):
        _capi.spmm(_capi.SUM, rp, col, val, X)
    planfree = min(wall(lambda: _capi.spmm(_capi.SUM, rp, col, val, X)) for _ in range(20))
    os.environ['DGS_PLAN'] = '0'  # the same PUBLIC operator with plans switched off: the like-for-like first-call yardstick
    A0 = fresh()
    dgsparse.spmm_sum(A0, X, 0)
    planfree_pub = sorted(wall(lambda: dgsparse.spmm_sum(A0, X, 0)) for _ in range(20))[10]
    os.environ['DGS_PLAN'] = '1'
    firsts = []
    for _ in range(10):
        A = fresh()
        firsts.append(wall(lambda: dgsparse.spmm_sum(A, X, 0)))
    first = sorted(firsts)[len(firsts) // 2]

    def loop(A, n=20):
        for _ in range(n):
            dgsparse.spmm_sum(A, X, 0)
    loops = []
    for _ in range(5):
        A = fresh()
        loops.append(wall(lambda: loop(A)))
    lazy20 = sorted(loops)[2]
    os.environ['DGS_PLAN'] = '0'
    A = fresh()
    loop(A)
    free20 = min(wall(lambda: loop(A)) for _ in range(5))
    os.environ['DGS_PLAN'] = '1'
    A = fresh()
    A.storage.spmm_plan('csr', N, wait=True)
    loop(A)
    planned20 = min(wall(lambda: loop(A)) for _ in range(5))
    A = fresh()
    eager20 = wall(lambda: (A.storage.spmm_plan('csr', N, wait=True), loop(A)))

    def loop_sync(A, n, every):  # a host synchronisation every `every` calls (an epoch's loss readback)
        for i in range(n):
            dgsparse.spmm_sum(A, X, 0)
            if (i + 1) % every == 0:
                torch.cuda.synchronize()
    A = fresh()
    lazy20s = wall(lambda: loop_sync(A, 20, 5))
    A = fresh()
    lazy200 = wall(lambda: loop(A, 200))
    os.environ['DGS_PLAN'] = '0'
    A = fresh()
    free200 = wall(lambda: loop(A, 200))
    os.environ['DGS_PLAN'] = '1'
    A = fresh()
    eager200 = wall(lambda: (A.storage.spmm_plan('csr', N, wait=True), loop(A, 200)))
print(json.dumps(dict(graph=name, feat=N, planfree_call_ms=round(planfree, 4), first_call_fresh_tensor_ms=round(first, 4),
                      planfree_public_op_ms=round(planfree_pub, 4), first_over_planfree_public_op=round(first / planfree_pub, 3),
                      first_over_planfree=round(first / planfree, 3), loop20_fresh_lazy_ms=round(lazy20, 3),
                      loop20_planfree_ms=round(free20, 3), loop20_plan_ready_ms=round(planned20, 3),
                      loop20_eager_blocking_build_ms=round(eager20, 3), loop20_fresh_lazy_sync_every_5_ms=round(lazy20s, 3),
                      loop200_fresh_lazy_ms=round(lazy200, 2), loop200_planfree_ms=round(free200, 2),
                      loop200_eager_blocking_build_ms=round(eager200, 2), plan_after=int(os.environ.get('DGS_PLAN_AFTER', '3')))))
