#!/usr/bin/env python3
"""Long-running randomized parity campaign (not collected by pytest): python tests/fuzz_gpu.py [seconds] [seed].
Random shapes / degree laws / value kinds / feature widths, every reduce + SDDMM + masked backward + csr2csc, each
checked against the CPU oracle with the same bars as tests/test_gpu_parity.py.  Prints one line per 50 cases."""
import faulthandler
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dgsparse-lib_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import oracle  # noqa: E402
from bench import graphgen  # noqa: E402
from dgsparse import _capi as capi  # noqa: E402
from util import around_matrix, assert_bitexact, assert_close, assert_sum_parity  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def one_case(rng, it):
    M = int(rng.choice([1, 3, 64, 65, 255, 256, 1000, 4096, 4097, 30000, 66000, 200000]))
    K = int(rng.choice([1, 2, 33, 1000, 20000, 150000]))
    N = int(rng.choice([1, 2, 3, 4, 8, 12, 16, 31, 32, 33, 64, 65, 100, 128, 256, 260]))
    per = float(rng.choice([0.0, 0.5, 3, 16, 60]))
    nnz = int(min(M * per, 1.5e6))
    alpha = float(rng.choice([1.5, 2.1, 3.0, 50.0]))
    dmax = int(rng.choice([4, 64, 300, 5000, 100000]))
    rp, col, st = graphgen.powerlaw_csr(M, max(nnz, 1), K=K, alpha=alpha, dmax=max(1, min(K * 3, dmax)), seed=it,
                                        dedup=bool(rng.integers(0, 2)), cols=str(rng.choice(['uniform', 'powerlaw'])))
    if rng.integers(0, 3) == 0 and col.shape[0]:  # unsorted columns in some rows
        col = col.copy()
        for r in rng.integers(0, M, 8):
            rng.shuffle(col[rp[r]:rp[r + 1]])
    kind = [None, 'tied', 'signed', 'uniform'][int(rng.integers(0, 4))]
    val = graphgen.weights(col.shape[0], kind, it) if kind else None
    X = (rng.integers(-3, 4, (K, N)) / 8).astype(np.float32)
    hubth = rng.choice(['', '1024', '2048', '0'])  # round 4: hub rows of the default sum / mean are chained above this length
    if hubth:
        os.environ['DGS_HUB_CHAIN'] = str(hubth)
    else:
        os.environ.pop('DGS_HUB_CHAIN', None)
    capi.reload_tuning()
    tag = f'it={it} M={M} K={K} N={N} nnz={col.shape[0]} maxdeg={st["max_deg"]} val={kind} hub={hubth or "default"}'
    if os.environ.get('FUZZ_VERBOSE'):
        print('case', tag, flush=True)
    drp, dcol, dval, dX = dev(rp), dev(col), (None if val is None else dev(val)), dev(X)
    C64 = oracle.spmm_sum_f64(rp, col, val, X)
    S64 = oracle.spmm_sum_f64(rp, col, val, X, absval=True)
    Emax = None
    for reduce in ('sum', 'mean', 'max', 'min'):
        C, E = capi.spmm(oracle.REDUCE[reduce], drp, dcol, dval, dX)
        C = C.cpu().numpy()
        Co, Eo = oracle.spmm(reduce, rp, col, val, X, fma=True)
        if reduce in ('max', 'min'):
            assert_bitexact(C, Co, tag + ' ' + reduce)
            assert_bitexact(E.cpu().numpy(), Eo, tag + ' E ' + reduce)
            if reduce == 'max':
                Emax = Eo
        else:
            sc = 1 if reduce == 'sum' else np.maximum(np.diff(rp), 1)[:, None]
            assert_sum_parity(C, Co, C64 / sc, S64 / sc, 1e-5, 2e-6, tag + ' ' + reduce, lens=np.diff(rp))
            # round 3: the strict-order schedule is bit-exact against its sequential chain for every row length
            for alg, fma in ((capi.ALG_STRICT_SUM, True), (capi.ALG_STRICT_NOFMA, False)):
                if rng.integers(0, 2) and not os.environ.get('FUZZ_NO_STRICT'):
                    Cs, _ = capi.spmm(oracle.REDUCE[reduce], drp, dcol, dval, dX, algorithm=alg)
                    assert_bitexact(Cs.cpu().numpy(), oracle.spmm(reduce, rp, col, val, X, fma=fma)[0], tag + f' strict {reduce} fma={fma}')
    if rng.integers(0, 2):  # round 3: fused epilogue == the unfused elementwise ops, bit for bit, on whatever schedule this shape takes
        red = ('sum', 'mean')[int(rng.integers(0, 2))]
        base, _ = capi.spmm(oracle.REDUCE[red], drp, dcol, dval, dX)
        bias = dev((rng.integers(-4, 5, N) / 8).astype(np.float32)) if rng.integers(0, 2) else None
        rs = dev((rng.integers(1, 5, M) / 4).astype(np.float32)) if rng.integers(0, 2) else None
        relu = bool(rng.integers(0, 2))
        got, _ = capi.spmm(oracle.REDUCE[red], drp, dcol, dval, dX, bias=bias, row_scale=rs, relu=relu)
        want = base
        if rs is not None:
            want = want * rs[:, None]
        if bias is not None:
            want = want + bias
        if relu:
            want = torch.relu(want)
        assert torch.equal(got, want), tag + f' epilogue {red} bias={bias is not None} rs={rs is not None} relu={relu}'
    capi.canary_check(tag + ' spmm')
    if col.shape[0] and M > 1 and rng.integers(0, 2) == 0:
        # round-2 entries: a forced locality plan must reproduce the plan-free results (max/min + E bit for bit, sum within
        # the bar), the accumulating sum adds into what C holds, the accumulating max merges a column split exactly
        plan = capi.spmm_plan(drp, dcol, K, N, force=True)
        if plan is not None:
            for reduce in ('sum', 'max', 'min'):
                C, E = capi.spmm(oracle.REDUCE[reduce], drp, dcol, dval, dX, plan=plan)
                Co, Eo = oracle.spmm(reduce, rp, col, val, X, fma=True)
                if reduce == 'sum':
                    assert_sum_parity(C.cpu().numpy(), Co, C64, S64, 1e-5, 2e-6, tag + ' plan sum', lens=np.diff(rp))
                else:
                    assert_bitexact(C.cpu().numpy(), Co, tag + ' plan ' + reduce)
                    assert_bitexact(E.cpu().numpy(), Eo, tag + ' plan E ' + reduce)
            if N % 4 == 0 and 32 <= N <= 256:  # round 3: SDDMM on the fused row-block / unit schedule over the same plan
                os.environ['DGS_SDDMM_FUSED'] = '1'
                capi.reload_tuning()
                D1p = (rng.integers(-3, 4, (M, N)) / 8).astype(np.float32)
                for mean in (False, True):
                    got = capi.sddmm(drp, dcol, dev(D1p), dX, reduce_op=capi.MEAN if mean else capi.SUM, plan=plan)
                    assert_close(got.cpu().numpy(), oracle.sddmm(rp, col, D1p, X, reduce='mean' if mean else 'sum', fma=True), 1e-5, 1e-5,
                                 tag + f' sddmm over the plan mean={mean}')
                os.environ.pop('DGS_SDDMM_FUSED', None)
                capi.reload_tuning()
        capi.canary_check(tag + ' plan')
        C0 = (rng.integers(-4, 5, (M, N)) / 4).astype(np.float32)
        Cacc = dev(C0)
        capi.spmm_acc(drp, dcol, dval, dX, Cacc, None, plan=plan)
        Co, _ = oracle.spmm('sum', rp, col, val, X, fma=True)
        live = np.diff(rp) > 0
        assert_sum_parity(Cacc.cpu().numpy()[live].astype(np.float64) - C0[live], Co[live], C64[live], S64[live] + np.abs(C0[live]), 1e-5, 4e-6,
                          tag + ' acc sum', lens=np.diff(rp)[live])
        assert_bitexact(Cacc.cpu().numpy()[~live], C0[~live], tag + ' acc sum leaves empty rows alone')
        srt = all(np.all(np.diff(col[rp[r]:rp[r + 1]]) >= 0) for r in range(M)) if M <= 5000 else False
        if srt and K >= 3:
            a, b = K // 3, K // 3 + max(1, K // 4)
            nl, h_lo = b - a, a
            ext = np.where((col >= a) & (col < b), col - a, np.where(col < a, nl + col, col)).astype(np.int32)
            Xe = np.concatenate([X[a:b], X[:a], X[b:]])
            Cu, Eu = oracle.spmm('max', rp, ext, val, Xe, fma=True)
            is_loc = (col >= a) & (col < b)
            rows_of = np.repeat(np.arange(M), np.diff(rp))
            vv = val if val is not None else np.ones(col.shape[0], np.float32)

            def sub(mask, shift, compact):
                cnt = np.bincount(rows_of[mask], minlength=M)
                keep = np.nonzero(cnt)[0] if compact else np.arange(M)
                rpp = np.concatenate([[0], np.cumsum(cnt[keep])]).astype(np.int32)
                return rpp, (ext[mask] - shift).astype(np.int32), vv[mask], keep.astype(np.int32)

            lrp, lcol, lval, _ = sub(is_loc, 0, False)
            rrp, rcol, rval, rrows = sub(~is_loc, nl, True)
            Xd = dev(Xe)
            if lcol.shape[0]:
                Cm, Em = capi.spmm(oracle.MAX, dev(lrp), dev(lcol), dev(lval), Xd[:nl].contiguous())
            else:
                Cm = torch.zeros((M, N), device='cuda')
                Em = torch.full((M, N), -1, dtype=torch.int32, device='cuda')
            if rcol.shape[0]:
                capi.spmm_acc_max(dev(rrp), dev(rcol), dev(rval), Xd[nl:].contiguous(), Cm, Em, dev(rrows), col_off=nl,
                                  n_local=nl, h_lo=h_lo)
            assert_bitexact(Cm.cpu().numpy(), Cu, tag + ' acc max values')
            assert_bitexact(Em.cpu().numpy(), Eu, tag + ' acc max E')
            # late round 3: the same cut for MIN - the lower columns folded in FRONT of the local result, the higher ones
            # BEHIND it (dgs_spmm_csr_acc_min_f32), features with signed zeros and, one case in three, NaN / inf entries that
            # the detector must find and the redo-only call must then repair (dgs_nonfinite_flag_f32, dgs_spmm_min_merge_f32)
            Xz = Xe.copy()
            Xz[Xz == 0] = rng.choice(np.array([0.0, -0.0], np.float32), size=int((Xz == 0).sum()))
            poisoned = rng.integers(0, 3) == 0
            if poisoned:
                for bad in (np.nan, np.inf, -np.inf):
                    Xz[rng.integers(0, K, 4), rng.integers(0, N, 4)] = bad
            Cn, En = oracle.spmm('min', rp, ext, val, Xz, fma=True)
            Xzd = dev(Xz)
            if lcol.shape[0]:
                Cq, Eq = capi.spmm(oracle.MIN, dev(lrp), dev(lcol), dev(lval), Xzd[:nl].contiguous())
            else:
                Cq = torch.zeros((M, N), device='cuda')
                Eq = torch.full((M, N), -1, dtype=torch.int32, device='cuda')
            halo = Xzd[nl:].contiguous()
            for mask, first in ((col < a, True), (col >= b, False)):
                srp, scol, sval, keep = sub(mask, nl, True)
                if scol.shape[0]:
                    capi.spmm_acc_min(dev(srp), dev(scol), dev(sval), halo, Cq, Eq, dev(keep), col_off=nl, precedes=first)
            flag = torch.zeros(1, dtype=torch.int32, device='cuda')
            capi.nonfinite_flag(Xzd, flag)
            capi.nonfinite_flag(dev(vv), flag)
            assert bool(int(flag)) == (not np.isfinite(Xz).all()), tag + ' nonfinite detector'
            if rrows.shape[0]:
                capi.spmm_min_merge(dev(rrows), None, None, None, 0, None, Cq, Eq, flag, drp, dev(ext), dev(vv), Xzd)
            assert_bitexact(Cq.cpu().numpy(), Cn, tag + f' acc min values (poisoned={poisoned})')
            assert_bitexact(Eq.cpu().numpy(), En, tag + f' acc min E (poisoned={poisoned})')
            # round 5: the same in ONE accumulating launch (dgs_spmm_csr_acc_min_around_f32): the local result a virtual entry
            # of its row; then the same redo-only call
            if lcol.shape[0]:
                Cq, Eq = capi.spmm(oracle.MIN, dev(lrp), dev(lcol), dev(lval), Xzd[:nl].contiguous())
            else:
                Cq = torch.zeros((M, N), device='cuda')
                Eq = torch.full((M, N), -1, dtype=torch.int32, device='cuda')
            arp, acol, aval, arows = around_matrix(rp, col, vv, np.diff(rp), a, b, M, compact=bool(rng.integers(0, 2)))
            if acol.shape[0]:
                darp, dacol = dev(arp), dev(acol)
                halo2 = halo if halo.shape[0] else torch.zeros((1, N), device='cuda')
                aplan = None
                if rng.integers(0, 2) and capi.spmm_schedule(oracle.MIN, arp.shape[0] - 1, halo2.shape[0] + M, N, acol.shape[0]) == 'rows':
                    aplan = capi.spmm_plan(darp, dacol, halo2.shape[0] + M, N, force=True)
                capi.spmm_acc_min_around(darp, dacol, dev(aval), halo2, Cq, Eq, dev(arows), nl, a, M, plan=aplan)
            if rrows.shape[0]:
                capi.spmm_min_merge(dev(rrows), None, None, None, 0, None, Cq, Eq, flag, drp, dev(ext), dev(vv), Xzd)
            assert_bitexact(Cq.cpu().numpy(), Cn, tag + f' acc min around values (poisoned={poisoned})')
            assert_bitexact(Eq.cpu().numpy(), En, tag + f' acc min around E (poisoned={poisoned})')
    if col.shape[0]:
        D1 = (rng.integers(-3, 4, (M, N)) / 8).astype(np.float32)
        dD1 = dev(D1)
        assert_close(capi.sddmm(drp, dcol, dD1, dX).cpu().numpy(), oracle.sddmm(rp, col, D1, X, fma=True), 1e-5, 1e-5,
                     tag + ' sddmm')
        assert_close(capi.sddmm(drp, dcol, dD1, dX, E=dev(Emax)).cpu().numpy(),
                     oracle.sddmm_mask(rp, col, D1, X, Emax, fma=True), 1e-5, 1e-5, tag + ' sddmm_mask')
        colptr, row, tval, perm = oracle.csr2csc(rp, col, val, K)
        gcolptr, grow, gval, gperm = capi.csr2csc(drp, dcol, dval, K)
        assert_bitexact(gcolptr.cpu().numpy(), colptr, tag + ' colptr')
        assert_bitexact(grow.cpu().numpy(), row, tag + ' cscrow')
        assert_bitexact(gperm.cpu().numpy(), perm, tag + ' perm')
        gX = capi.spmm_mask(gcolptr, grow, gval, dD1, dev(Emax)).cpu().numpy()
        ref = oracle.spmm_mask(colptr, row, tval, D1, Emax, fma=True)
        assert_sum_parity(gX, ref, oracle.spmm_mask_f64(colptr, row, tval, D1, Emax),
                          oracle.spmm_mask_f64(colptr, row, tval, D1, Emax, absval=True), 1e-5, 2e-6, tag + ' spmm_mask',
                          lens=np.diff(colptr))
        # the single-pass backward (scatter over the arg ids) must give the same two gradients
        aX, aW = capi.spmm_arg_backward(drp, dcol, dval, dev(Emax), dD1, dX)
        tv = tval if val is not None else np.ones_like(tval)
        assert_sum_parity(aX.cpu().numpy(), ref, oracle.spmm_mask_f64(colptr, row, tv, D1, Emax),
                          oracle.spmm_mask_f64(colptr, row, tv, D1, Emax, absval=True), 1e-5, 2e-6,
                          tag + ' arg_backward gX', lens=np.diff(colptr))
        if val is not None:
            assert_close(aW.cpu().numpy(), oracle.sddmm_mask(rp, col, D1, X, Emax, fma=True), 1e-5, 1e-5,
                         tag + ' arg_backward gW')
    capi.canary_check(tag + ' sddmm / csr2csc / backward')
    return tag


def main():
    faulthandler.enable(all_threads=True)  # a native crash prints the Python stack of every thread before the core dump
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    t0, it = time.time(), 0
    while time.time() - t0 < budget:
        tag = one_case(rng, seed * 100000 + it)
        it += 1
        if it % 50 == 0:
            print(f'{it} cases ok, {time.time() - t0:.0f} s, last: {tag}', flush=True)
    print(f'FUZZ OK: {it} cases in {time.time() - t0:.0f} s (seed {seed}, canaries {"on" if capi._CANARY else "off"})')


if __name__ == '__main__':
    main()
