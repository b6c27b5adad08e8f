#!/usr/bin/env python3
"""Randomized parity campaign on the CPU EMULATION of the kernels (tests/emu; not collected by pytest):
    python tests/fuzz_emu.py [seconds] [seed]
Random shapes / degree laws / value kinds / feature widths / hub thresholds; every reduce plan-free and over a forced plan, the
strict-order modes, the fused epilogue - each against the oracle with the bars of tests/test_gpu_parity.py.  The GPU twin is
tests/fuzz_gpu.py; this one needs no GPU and additionally turns barrier-protocol bugs into deadlock reports.
Relaxed-memory campaign (round 6): DGS_EMU_MEM=relaxed DGS_EMU_BLOCKS=24 DGS_EMU_BLOCK_ORDER=rand:<s> DGS_EMU_PREEMPT=40 FUZZ_FOLD=1
- every case folds its partial rows in the kernel, the emulator's memory model serves stale data to a broken hand-over, and every
case must end with ZERO hazards on its counters."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'emu')):
    sys.path.insert(0, p)
import emu_lib as E  # noqa: E402
import oracle  # noqa: E402
from bench import graphgen  # noqa: E402
from util import around_matrix, assert_bitexact, assert_sum_parity  # noqa: E402

OPS = {'sum': E.SUM, 'max': E.MAX, 'min': E.MIN, 'mean': E.MEAN}


def one_case(rng, it):
    M = int(rng.choice([1, 3, 64, 65, 255, 1000, 4096, 4097, 30000, 66000, 70000]))
    K = int(rng.choice([1, 2, 33, 1000, 20000]))
    N = int(rng.choice([1, 2, 3, 4, 8, 12, 16, 31, 32, 33, 64, 65, 100, 128, 256, 260]))
    per = float(rng.choice([0.0, 0.5, 3, 16, 70]))
    nnz = int(min(M * per, 3.0e5))
    alpha = float(rng.choice([1.5, 2.1, 3.0, 50.0]))
    dmax = int(rng.choice([4, 64, 300, 3000, 9000]))
    if rng.integers(0, 2):  # half of the cases: the general schedule with hub rows (what round 4 added)
        M, per, alpha = int(rng.choice([66000, 70000])), float(rng.choice([0.5, 3])), 1.5
        K, dmax = int(rng.choice([1000, 20000])), int(rng.choice([3000, 9000]))
        N = int(rng.choice([16, 20, 32, 48, 64, 100, 128, 256, 41, 7, 8, 4, 1, 33]))
        nnz = int(M * per)
    rp, col, st = graphgen.powerlaw_csr(M, max(nnz, 1), K=K, alpha=alpha, dmax=max(1, min(K * 3, dmax)), seed=it,
                                        dedup=bool(rng.integers(0, 2)), cols=str(rng.choice(['uniform', 'powerlaw'])))
    tiny_hub = M in (1000, 4096, 4097, 30000) and K >= 1000 and rng.integers(0, 2)  # single-launch inputs with hub rows
    if (M >= 66000 and K >= 33) or tiny_hub:  # the generator rescales its degrees to the nnz budget: put real hub rows in by hand
        lens0 = np.diff(rp).astype(np.int64)
        rows = rng.choice(M, int(rng.integers(2, 7)), replace=False)
        parts = np.split(col, rp[1:-1])
        for r in rows:
            d = int(rng.choice([1025, 1030, 1500, 2049, 3000, 5000, 9000]))
            parts[r] = np.sort(rng.choice(K, d, replace=d > K)).astype(np.int32)
            lens0[r] = d
        col = np.concatenate(parts).astype(np.int32)
        rp = np.zeros(M + 1, np.int32)
        rp[1:] = np.cumsum(lens0)
        st = dict(st, max_deg=int(lens0.max()))
    shuffled = False
    if rng.integers(0, 3) == 0 and col.shape[0]:
        shuffled = True
        col = col.copy()
        for r in rng.integers(0, M, 8):
            rng.shuffle(col[rp[r]:rp[r + 1]])
    kind = [None, 'tied', 'signed', 'uniform'][int(rng.integers(0, 4))]
    val = graphgen.weights(col.shape[0], kind, it) if kind else None
    X = (rng.integers(-3, 4, (K, N)) / 8).astype(np.float32)
    hubth = int(rng.choice([0, 1024, 1024, 2048, 16384]))
    fold = int(rng.integers(0, 2))  # round 5: partial rows folded inside the fused launch (last-arriving unit wave) / by the combine launch
    if os.environ.get('FUZZ_FOLD'):  # relaxed-memory campaigns: every case folds in the kernel (that IS the hand-over under test)
        fold = int(os.environ['FUZZ_FOLD'])
    E.set_env(DGS_HUB_CHAIN=hubth, DGS_NBU=int(rng.choice([8, 16, 64])), DGS_STRICT_NBU=int(rng.choice([8, 16, 64])), DGS_FOLD=fold)
    # round 5: callers that know the longest row say so (DGS_ALG_NO_HUB_ROWS) - the kernels without the hub role, same results
    hint = E.ALG_NO_HUB_ROWS if (hubth > 0 and int(np.diff(rp).max(initial=0)) <= max(hubth, 1024) and rng.integers(0, 2)) else 0
    tag = f'it={it} M={M} K={K} N={N} nnz={col.shape[0]} maxdeg={st["max_deg"]} val={kind} hub={hubth} fold={fold} hint={hint:#x}'
    if os.environ.get('FUZZ_VERBOSE'):
        print('case', tag, flush=True)
    if os.environ.get('FUZZ_DUMP'):  # the inputs of the case that is about to run (to replay a hang or a crash outside the campaign)
        np.savez(os.environ['FUZZ_DUMP'], rp=rp, col=col, val=np.zeros(0, np.float32) if val is None else val, X=X, K=K, hubth=hubth)
    lens = np.diff(rp)
    C64 = oracle.spmm_sum_f64(rp, col, val, X)
    S64 = oracle.spmm_sum_f64(rp, col, val, X, absval=True)
    sched = E.schedule(E.SUM, M, K, N, max(col.shape[0], 0)) if col.shape[0] else 'small'
    plan = E.spmm_plan(rp, col, K) if (sched == 'rows' and rng.integers(0, 2)) else None
    hub_ok = hubth > 0 and sched != 'panel'  # every feature width, general and single-launch schedules (panel: its own long-row bound)
    for reduce in ('sum', 'mean', 'max', 'min'):
        Co, Eo = oracle.spmm(reduce, rp, col, val, X, fma=True)
        for kw in ([{}] + ([dict(plan=plan)] if plan is not None else [])):
            C, Ee = E.spmm(OPS[reduce], rp, col, val, X, **(kw if kw else dict(algorithm=hint)))
            what = f'{tag} {reduce} {"plan" if kw else "plan-free"}'
            if reduce in ('max', 'min'):
                assert_bitexact(C, Co, what)
                assert_bitexact(Ee, Eo, what + ' E')
            else:
                sc = 1 if reduce == 'sum' else np.maximum(lens, 1)[:, None]
                assert_sum_parity(C, Co, C64 / sc, S64 / sc, 1e-5, 2e-6, what, lens=lens)
                if hub_ok and (lens > max(hubth, 1024)).any():
                    hub = lens > max(hubth, 1024)
                    assert_bitexact(C[hub], Co[hub], what + ' hub rows')
        if reduce in ('sum', 'mean'):
            for alg, fma in ((E.ALG_STRICT_SUM, True), (E.ALG_STRICT_NOFMA, False)):
                if rng.integers(0, 2):
                    Cs, _ = E.spmm(OPS[reduce], rp, col, val, X, algorithm=alg)
                    assert_bitexact(Cs, oracle.spmm(reduce, rp, col, val, X, fma=fma)[0], f'{tag} strict {reduce} fma={fma}')
                    if plan is not None:  # round 5: the same chains over the plan's strict table (one launch)
                        Cp = E.spmm_ex(OPS[reduce], rp, col, val, X, algorithm=alg, plan=plan)
                        assert_bitexact(Cp, Cs, f'{tag} strict over the plan {reduce} fma={fma}')
    if not shuffled and K >= 2 and col.shape[0] and val is not None and rng.integers(0, 2):
        # round 5: the multi-GPU min's halo part in ONE accumulating launch (dgs_spmm_csr_acc_min_around_f32): columns cut in
        # [lower | local | higher] at random, the local result a virtual entry of its row; == algorithm 0 on the undivided row
        a = int(rng.integers(0, K))
        b = int(rng.integers(a, K + 1))
        nl = b - a
        ext = np.where((col >= a) & (col < b), col - a, np.where(col < a, nl + col, col)).astype(np.int32)
        Xe = np.ascontiguousarray(np.concatenate([X[a:b], X[:a], X[b:]]))
        Co, Eo = oracle.spmm('min', rp, ext, val, Xe, fma=True)
        is_loc = (col >= a) & (col < b)
        rows_of = np.repeat(np.arange(M), lens)
        lrp = np.concatenate([[0], np.cumsum(np.bincount(rows_of[is_loc], minlength=M))]).astype(np.int32)
        if nl and is_loc.any():
            C, Ee = E.spmm(E.MIN, lrp, np.ascontiguousarray(ext[is_loc]), np.ascontiguousarray(val[is_loc]), np.ascontiguousarray(Xe[:nl]))
        else:
            C, Ee = np.zeros((M, N), np.float32), np.full((M, N), -1, np.int32)
        arp, acol, aval, arows = around_matrix(rp, col, val, lens, a, b, M, compact=bool(rng.integers(0, 2)))
        if acol.size:
            Xh = np.ascontiguousarray(Xe[nl:]) if K - nl else np.zeros((1, N), np.float32)
            aplan = None
            if E.schedule(E.MIN, arp.size - 1, Xh.shape[0] + M, N, acol.size) == 'rows' and rng.integers(0, 2):
                aplan = E.spmm_plan(arp, acol, Xh.shape[0] + M)
            E.spmm_acc_min_around(arp, acol, aval, Xh, C, Ee, arows, nl, a, M, plan=aplan)
        assert_bitexact(C, Co, f'{tag} min around [{a}, {b}) values')
        assert_bitexact(Ee, Eo, f'{tag} min around [{a}, {b}) E')
    if rng.integers(0, 2):
        red = ('sum', 'mean')[int(rng.integers(0, 2))]
        kw = dict(plan=plan) if (plan is not None and rng.integers(0, 2)) else {}
        base, _ = E.spmm(OPS[red], rp, col, val, X, **kw)
        bias = (rng.integers(-4, 5, N) / 8).astype(np.float32) if rng.integers(0, 2) else None
        rs = (rng.integers(1, 5, M) / 4).astype(np.float32) if rng.integers(0, 2) else None
        relu = bool(rng.integers(0, 2))
        got = E.spmm_ex(OPS[red], rp, col, val, X, bias=bias, row_scale=rs, relu=relu, **kw)
        want = base
        if rs is not None:
            want = want * rs[:, None]
        if bias is not None:
            want = want + bias[None, :]
        if relu:
            want = np.where(want < 0, np.float32(0), want)
        assert_bitexact(got, want.astype(np.float32), tag + f' epilogue {red}')
    if os.environ.get('DGS_EMU_MEM') == 'relaxed':  # the memory model's hazard counters (tests/emu/emu_rt.cpp): a correct hand-over has none
        rep = E.mem_report()
        assert rep['unperformed'] == 0 and rep['l1_stale'] == 0, f'{tag}: memory-model hazards {rep}'
        one_case.mem_ops = getattr(one_case, 'mem_ops', 0) + rep['loads'] + rep['stores']
    return tag


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0, n = time.time(), 0
    while time.time() - t0 < budget:
        tag = one_case(rng, seed * 100000 + n)
        n += 1
        if n % 20 == 0:
            print(f'{n} cases green, {time.time() - t0:.0f} s (last: {tag})', flush=True)
    extra = f", relaxed-memory mode: {getattr(one_case, 'mem_ops', 0)} modelled accesses, 0 hazards" if os.environ.get('DGS_EMU_MEM') == 'relaxed' else ''
    print(f'done: {n} cases green in {time.time() - t0:.0f} s, seed {seed}' + extra)


if __name__ == '__main__':
    main()
