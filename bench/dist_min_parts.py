"""Compute-side cost of the overlapped min of dgsparse.dist on ONE rank's shard (no exchange: standalone plan, halo rows
filled with random features): the one-pass product that has to wait for the exchange, against the pieces of the overlapped
schedule - local product (runs under the exchange), detector scans, the two accumulating halo products (or, round 5, the ONE
launch in which the local result is a virtual entry of its row), the redo-only call.
    python bench/dist_min_parts.py [rows_log2=20] [deg=16] [feat=64] [world=8] [locality=0.8]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dgsparse-lib_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def ms(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    from dgsparse import dist as dd
    av = [float(x) for x in sys.argv[1:]]
    rows_log2, deg, N, world = (int(av[i]) if len(av) > i else d for i, d in enumerate((20, 16, 64, 8)))
    locality = av[4] if len(av) > 4 else 0.8
    dev = torch.device('cuda:0')
    part = dd.synthetic_partition(3 % world, world, 1 << rows_log2, deg, locality=locality, seed=0, device=dev)
    eng = dd.DistSpMM(part, N, standalone=True)
    p, plan, ops = eng.part, eng.plan, eng.ops
    eng.B_ext.copy_(torch.rand(eng.B_ext.shape, device=dev))
    nl = p.n_local
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    R = int(plan.rem_rows.numel())
    print(f'shard: {nl} rows, {p.nnz} nnz, {eng.n_halo} halo rows, {R} rows with remote entries, '
          f'{int(plan.rem[1].numel())} remote nnz, feat {N}')
    one = ms(lambda: eng.compute('min'))
    loc = ms(lambda: ops.spmm(2, plan.loc[0], plan.loc[1], plan.loc[2], eng.B_ext[:nl], shared_gpu=True))
    scan_loc = ms(lambda: (ops.nonfinite_flag(eng.B_ext[:nl], flag), ops.nonfinite_flag(p.val, flag)))
    scan_halo = ms(lambda: ops.nonfinite_flag(eng.B_ext[nl:], flag))
    C, E = ops.spmm(2, plan.loc[0], plan.loc[1], plan.loc[2], eng.B_ext[:nl])
    halo = eng.B_ext[nl:]
    (lo, lo_rows, _), (hi, hi_rows, _) = plan.min_parts()
    acc_lo = ms(lambda: ops.spmm_acc_min(lo[0], lo[1], lo[2], halo, C, E, lo_rows, nl, True))
    acc_hi = ms(lambda: ops.spmm_acc_min(hi[0], hi[1], hi[2], halo, C, E, hi_rows, nl, False))
    # round 5: both sides in ONE accumulating launch, the local result a virtual entry of its row (dgs_spmm_csr_acc_min_around_f32).
    # Repeated in place like the two folds above (a repeat re-folds a finished row: same work, the values no longer matter)
    ar, ar_rows, _ = plan.min_around()
    acc_around = ms(lambda: ops.spmm_acc_min_around(ar[0], ar[1], ar[2], halo, C, E, ar_rows, nl, plan.h_lo, nl))
    flag.zero_()
    redo = ms(lambda: ops.min_redo(plan.rem_rows, C, E, flag, p.rowptr, plan.col_ext, p.val, eng.B_ext))
    relabel = ms(lambda: ops.relabel(E, plan.ext2glob32))
    after = scan_halo + acc_lo + acc_hi + redo
    print(f'one pass after the exchange                      {one:8.4f} ms')
    print(f'under the exchange: local min {loc:.4f} + scans of local features / values {scan_loc:.4f}')
    print(f'after the exchange: scan of the halo {scan_halo:.4f} + lower halo folded in front {acc_lo:.4f} '
          f'({int(lo[1].numel())} nnz, {int(lo_rows.numel())} rows) + higher halo behind {acc_hi:.4f} '
          f'({int(hi[1].numel())} nnz, {int(hi_rows.numel())} rows) + redo-if-flagged {redo:.4f} = {after:8.4f} ms')
    print(f'exposed after the exchange: {one:.4f} -> {after:.4f} ms; total GPU work {one:.4f} -> {loc + scan_loc + after:.4f} ms')
    after1 = scan_halo + acc_around + redo
    print(f'ONE accumulating launch (min_form around, the default): scan of the halo {scan_halo:.4f} + halo around the local result '
          f'{acc_around:.4f} ({int(ar[1].numel())} entries incl. {int(ar[1].numel()) - int(lo[1].numel()) - int(hi[1].numel())} virtual, '
          f'{int(ar_rows.numel())} rows) + redo-if-flagged {redo:.4f} = {after1:8.4f} ms exposed (two launches: {after:.4f})')
    print(f'global column ids of E on demand (DistSpMM.last_E): {relabel:.4f} ms')


if __name__ == '__main__':
    main()
