"""Strict-order sum / mean (DGS_ALG_STRICT_SUM / DGS_ALG_STRICT_NOFMA): every (row, feature) is ONE sequential chain in
CSR order, i.e. literally algorithm 0 (/root/reference/include/cuda/spmm_cuda.cuh:27-47; host twin
example/util/sp_util.hpp:73-83).  Bar: BIT-EXACT against the oracle's sequential chain for every row length and every
feature mapping - fmaf chain for STRICT_SUM, separately rounded multiply and add for STRICT_NOFMA (the latter is also
what the reference's own compiled host loop, oracle/_ref, produces)."""
import os

import numpy as np
import pytest
import torch

import oracle
from bench import graphgen
from util import assert_bitexact

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def capi():
    from dgsparse import _capi
    return _capi


def run(capi, op, rp, col, val, X, alg):
    d = 'cuda'
    C, _ = capi.spmm({'sum': capi.SUM, 'mean': capi.MEAN}[op], torch.from_numpy(rp).to(d), torch.from_numpy(col).to(d),
                     None if val is None else torch.from_numpy(val).to(d), torch.from_numpy(X).to(d), algorithm=alg)
    return C.cpu().numpy()


def hub_graph(M=24000, nnz=700000, dmax=20000, seed=3, cols='powerlaw'):
    rp, col, st = graphgen.powerlaw_csr(M, nnz, alpha=1.9, dmax=dmax, seed=seed, cols=cols)
    deg = np.diff(rp)
    # every row class of the strict schedule must be present: short, whole-tile units, 4-slice units, hub units
    assert (deg <= 64).any() and ((deg > 64) & (deg <= 256)).any() and ((deg > 256) & (deg <= 2048)).any() and (deg > 2048).any()
    return rp, col, st


@pytest.mark.parametrize('N', [1, 3, 4, 8, 16, 20, 32, 33, 64, 128, 256, 384])
@pytest.mark.parametrize('mode', ['fma', 'nofma'])
def test_strict_sum_bitexact_every_feature_mapping(capi, N, mode):
    rp, col, st = hub_graph()
    val = graphgen.weights(col.shape[0], 'signed', 5)
    X = graphgen.features(st['K'], N, 6) - 0.3
    alg = capi.ALG_STRICT_SUM if mode == 'fma' else capi.ALG_STRICT_NOFMA
    C = run(capi, 'sum', rp, col, val, X, alg)
    ref, _ = oracle.spmm('sum', rp, col, val, X, fma=(mode == 'fma'), threads=oracle.max_threads())
    assert_bitexact(C, ref, f'strict sum N={N} {mode}')


@pytest.mark.parametrize('op,has_val', [('mean', True), ('sum', False), ('mean', False)])
@pytest.mark.parametrize('N', [32, 64, 100])
def test_strict_mean_and_unit_weights(capi, op, has_val, N):
    rp, col, st = hub_graph(seed=4)
    val = graphgen.weights(col.shape[0], 'uniform', 7) if has_val else None
    X = graphgen.features(st['K'], N, 8)
    for alg, fma in ((capi.ALG_STRICT_SUM, True), (capi.ALG_STRICT_NOFMA, False)):
        C = run(capi, op, rp, col, val, X, alg)
        ref, _ = oracle.spmm(op, rp, col, val, X, fma=fma, threads=oracle.max_threads())
        assert_bitexact(C, ref, f'strict {op} N={N} val={has_val} fma={fma}')


def test_strict_matches_the_reference_host_loop(capi):
    """NOFMA mode against the reference's own spmm_reference_host compiled by g++ (oracle/_ref), all rows."""
    if not oracle.have_ref():
        pytest.skip('oracle/_ref not built')
    rp, col, st = hub_graph(seed=9)
    val = graphgen.weights(col.shape[0], 'uniform', 1)
    X = graphgen.features(st['K'], 64, 2)
    C = run(capi, 'sum', rp, col, val, X, capi.ALG_STRICT_NOFMA)
    assert_bitexact(C, oracle.ref_spmm_sum(rp, col, val, X), 'strict nofma vs spmm_reference_host')


def test_strict_small_single_launch_path(capi):
    """Inputs that take the single-launch schedule: long rows are whole-tile strict units inside the row blocks."""
    rp, col, st = graphgen.powerlaw_csr(3000, 60000, alpha=1.8, dmax=2500, seed=11)
    assert np.diff(rp).max() > 300
    val = graphgen.weights(col.shape[0], 'signed', 2)
    for N in (4, 16, 64, 65, 128):
        X = graphgen.features(st['K'], N, 3) - 0.5
        for op in ('sum', 'mean'):
            for alg, fma in ((capi.ALG_STRICT_SUM, True), (capi.ALG_STRICT_NOFMA, False)):
                C = run(capi, op, rp, col, val, X, alg)
                ref, _ = oracle.spmm(op, rp, col, val, X, fma=fma)
                assert_bitexact(C, ref, f'strict small {op} N={N} fma={fma}')


def test_strict_on_the_column_panel_schedule(capi):
    """Dense graph forced onto the column-panel sweep: rows up to tlong are sequential fmaf chains there already, the
    longer ones go through strict unit waves (no combine)."""
    rp, col, st = graphgen.powerlaw_csr(9000, 2_400_000, alpha=2.6, dmax=8000, seed=5)
    val = graphgen.weights(col.shape[0], 'uniform', 3)
    X = graphgen.features(st['K'], 64, 4)
    old = {k: os.environ.get(k) for k in ('DGS_PANEL', 'DGS_PANEL_TLONG')}
    os.environ['DGS_PANEL'] = '1'
    os.environ['DGS_PANEL_TLONG'] = '2500'
    capi.reload_tuning()
    try:
        assert np.diff(rp).max() > 2500
        C = run(capi, 'sum', rp, col, val, X, capi.ALG_STRICT_SUM)
        Cn = run(capi, 'mean', rp, col, val, X, capi.ALG_STRICT_NOFMA)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        capi.reload_tuning()
    ref, _ = oracle.spmm('sum', rp, col, val, X, fma=True, threads=oracle.max_threads())
    assert_bitexact(C, ref, 'strict sum on the panel schedule')
    refn, _ = oracle.spmm('mean', rp, col, val, X, fma=False, threads=oracle.max_threads())
    assert_bitexact(Cn, refn, 'strict nofma mean (never the panel schedule)')


def test_strict_is_deterministic_and_ignored_by_max(capi):
    rp, col, st = hub_graph(seed=12)
    val = graphgen.weights(col.shape[0], 'tied', 1)
    X = graphgen.features(st['K'], 64, 2)
    a = run(capi, 'sum', rp, col, val, X, capi.ALG_STRICT_SUM)
    b = run(capi, 'sum', rp, col, val, X, capi.ALG_STRICT_SUM)
    assert_bitexact(a, b, 'strict run to run')
    d = 'cuda'
    args = (torch.from_numpy(rp).to(d), torch.from_numpy(col).to(d), torch.from_numpy(val).to(d), torch.from_numpy(X).to(d))
    C0, E0 = capi.spmm(capi.MAX, *args)
    C1, E1 = capi.spmm(capi.MAX, *args, algorithm=capi.ALG_STRICT_SUM)
    assert torch.equal(C0, C1) and torch.equal(E0, E1)


@pytest.mark.parametrize('N', [64, 128])
def test_strict_headline_graph_all_rows(capi, N):
    """The bench workload at full size (2^20 rows, ~16 nnz/row, max degree ~5e4): EVERY element bit-exact against the
    sequential chain, so parity.within_1e_5 of `bench.py --strict` is true by construction."""
    rp, col, st = graphgen.dataset_shaped('synth1m', seed=0, device='cuda', as_torch=True)
    g = torch.Generator(device='cuda')
    g.manual_seed(1)
    val = torch.rand(st['nnz'], generator=g, device='cuda')
    X = torch.rand((st['K'], N), generator=g, device='cuda')
    rpc, colc, valc, Xc = rp.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy(), X.cpu().numpy()
    for alg, fma in ((capi.ALG_STRICT_SUM, True), (capi.ALG_STRICT_NOFMA, False)):
        C, _ = capi.spmm(capi.SUM, rp, col, val, X, algorithm=alg)
        ref, _ = oracle.spmm('sum', rpc, colc, valc, Xc, fma=fma, threads=oracle.max_threads())
        assert_bitexact(C.cpu().numpy(), ref, f'headline graph strict fma={fma} N={N}')


# ---- round 4: the DEFAULT sum / mean chain the hub rows (longer than DGS_HUB_CHAIN nnz) -----------------------------------
def _default(capi, op, rp, col, val, X, plan=False, **kw):
    d = 'cuda'
    drp, dcol = torch.from_numpy(rp).to(d), torch.from_numpy(col).to(d)
    dX = torch.from_numpy(X).to(d)
    pl = capi.spmm_plan(drp, dcol, X.shape[0], X.shape[1], force=True) if plan else None
    C, _ = capi.spmm({'sum': capi.SUM, 'mean': capi.MEAN}[op], drp, dcol, None if val is None else torch.from_numpy(val).to(d),
                     dX, plan=pl, **kw)
    torch.cuda.synchronize()
    return C.cpu().numpy(), pl


@pytest.mark.first_contact
@pytest.mark.parametrize('N', [16, 32, 64, 128, 256, 48, 384, 41, 100, 8, 4, 7, 1])
@pytest.mark.parametrize('plan', [False, True], ids=['plan-free', 'plan'])
def test_default_sum_chains_the_hub_rows(capi, monkeypatch, N, plan):
    """Rows above the hub threshold are ONE fmaf chain per feature in the default schedule too - bit for bit the oracle's
    sequential chain, like the rows up to 64 nnz; the rows in between keep the fixed tree (1e-5).  Plan-free (hub table from
    the classify pass) and planned (hub table of the plan, the hub rows' units skipped), every 16-byte feature mapping,
    widths that leave a slice partly or wholly empty (48), several feature tiles (384), scalar lanes (41, 7, 1) and the narrow
    tiles whose hub workgroup runs the shorter gather sets (4, 8)."""
    monkeypatch.setenv('DGS_HUB_CHAIN', '1024')
    rp, col, st = graphgen.powerlaw_csr(70000, 900000, alpha=1.9, dmax=20000, seed=21)
    lens = np.diff(rp)
    hub = lens > 1024
    assert hub.sum() >= 8 and lens.max() > 5000  # (the generator rescales its degrees to the nnz budget: the longest row is ~7 500)
    val = graphgen.weights(col.shape[0], 'uniform', 5)
    X = graphgen.features(st['K'], N, 6)
    ref, _ = oracle.spmm('sum', rp, col, val, X, fma=True, threads=oracle.max_threads())
    C, pl = _default(capi, 'sum', rp, col, val, X, plan)
    if plan:
        assert pl is not None and pl.info.n_hub == int(hub.sum())
    assert_bitexact(C[hub], ref[hub], f'hub rows N={N}')
    assert_bitexact(C[lens <= 64], ref[lens <= 64], f'short rows N={N}')
    rel = np.abs(C - ref) / np.maximum(np.abs(ref), 1e-6)
    assert rel.max() <= 1e-5
    # mean and unit weights through the same path
    Cm, _ = _default(capi, 'mean', rp, col, None, X, plan)
    refm, _ = oracle.spmm('mean', rp, col, None, X, fma=True, threads=oracle.max_threads())
    assert_bitexact(Cm[hub], refm[hub], f'hub rows mean, unit weights N={N}')
    assert np.abs(Cm - refm).max() <= 1e-5 * np.abs(refm).max() + 1e-6


@pytest.mark.first_contact
def test_hub_chain_switch_and_other_reduces_share_the_plan(capi, monkeypatch):
    """DGS_HUB_CHAIN=0 brings the fixed tree back for every row; a plan built WITH hub rows still serves max / min (their
    units stay in the table, sorted behind the other units of each XCD's share) bit-exactly."""
    rp, col, st = graphgen.powerlaw_csr(70000, 900000, alpha=1.9, dmax=20000, seed=22)
    lens = np.diff(rp)
    val = graphgen.weights(col.shape[0], 'signed', 7)
    X = graphgen.features(st['K'], 64, 8) - 0.4
    monkeypatch.setenv('DGS_HUB_CHAIN', '2048')
    d = 'cuda'
    drp, dcol, dval, dX = (torch.from_numpy(a).to(d) for a in (rp, col, val, X))
    pl = capi.spmm_plan(drp, dcol, st['K'], 64, force=True)
    assert pl.info.n_hub == int((lens > 2048).sum()) > 0
    for red in ('max', 'min'):
        C, E = capi.spmm(oracle.REDUCE[red], drp, dcol, dval, dX, plan=pl)
        Co, Eo = oracle.spmm(red, rp, col, val, X, fma=True)
        assert_bitexact(C.cpu().numpy(), Co, red + ' over a plan with hub rows')
        assert_bitexact(E.cpu().numpy(), Eo, red + ' E over a plan with hub rows')
    Chub, _ = capi.spmm(capi.SUM, drp, dcol, dval, dX, plan=pl)
    monkeypatch.setenv('DGS_HUB_CHAIN', '0')
    Ctree, _ = capi.spmm(capi.SUM, drp, dcol, dval, dX, plan=pl)        # same plan, hub blocks off: units of the hub rows walked
    Cfree, _ = capi.spmm(capi.SUM, drp, dcol, dval, dX)
    ref, _ = oracle.spmm('sum', rp, col, val, X, fma=True, threads=oracle.max_threads())
    hub = lens > 2048
    assert_bitexact(Chub.cpu().numpy()[hub], ref[hub], 'hub rows chained')
    for name, Cx in (('planned tree', Ctree), ('plan-free tree', Cfree)):
        Cx = Cx.cpu().numpy()
        assert not np.array_equal(Cx[hub].view(np.int32), ref[hub].view(np.int32)), name + ': the switch must bring the tree back'
        C64 = oracle.spmm_sum_f64(rp, col, val, X)
        S64 = oracle.spmm_sum_f64(rp, col, val, X, absval=True)
        assert (np.abs(Cx - C64) <= 3e-6 * S64 + 1e-6).all(), name


@pytest.mark.first_contact
def test_hub_chain_with_the_fused_epilogue_and_the_panel_schedule(capi, monkeypatch):
    """The epilogue leaves with the hub rows' results as well (bit-identical to the unfused ops), and a dense graph on the
    column-panel sweep chains its rows above the hub threshold."""
    monkeypatch.setenv('DGS_HUB_CHAIN', '1024')
    rp, col, st = graphgen.powerlaw_csr(70000, 900000, alpha=1.9, dmax=20000, seed=23)
    lens = np.diff(rp)
    hub = lens > 1024
    val = graphgen.weights(col.shape[0], 'uniform', 3)
    X = graphgen.features(st['K'], 64, 4)
    d = 'cuda'
    drp, dcol, dval, dX = (torch.from_numpy(a).to(d) for a in (rp, col, val, X))
    bias = torch.linspace(-1, 1, 64, device=d)
    rs = torch.rand(rp.size - 1, device=d) + 0.5
    C0, _ = capi.spmm(capi.SUM, drp, dcol, dval, dX)
    want = torch.relu(C0 * rs[:, None] + bias)
    for pl in (None, capi.spmm_plan(drp, dcol, st['K'], 64, force=True)):
        got, _ = capi.spmm(capi.SUM, drp, dcol, dval, dX, plan=pl, bias=bias, row_scale=rs, relu=True)
        base, _ = capi.spmm(capi.SUM, drp, dcol, dval, dX, plan=pl)
        assert torch.equal(got, torch.relu(base * rs[:, None] + bias)), 'fused == unfused bit for bit (hub rows included)'
        assert torch.equal(got[torch.from_numpy(hub).to(d)], want[torch.from_numpy(hub).to(d)]), 'hub rows do not depend on the plan'
    # dense graph forced onto the panel schedule: rows above max(tlong, hub threshold) are chains
    rp2, col2, st2 = graphgen.powerlaw_csr(9000, 2_400_000, alpha=2.6, dmax=8000, seed=5)
    val2 = graphgen.weights(col2.shape[0], 'uniform', 3)
    X2 = graphgen.features(st2['K'], 64, 4)
    monkeypatch.setenv('DGS_PANEL', '1')
    monkeypatch.setenv('DGS_PANEL_TLONG', '2500')
    monkeypatch.setenv('DGS_HUB_CHAIN', '3000')
    assert capi.spmm_schedule(capi.SUM, rp2.size - 1, st2['K'], 64, col2.size) == 'panel'
    C2, _ = _default(capi, 'sum', rp2, col2, val2, X2)
    ref2, _ = oracle.spmm('sum', rp2, col2, val2, X2, fma=True, threads=oracle.max_threads())
    l2 = np.diff(rp2)
    assert (l2 > 3000).any()
    assert_bitexact(C2[l2 > 3000], ref2[l2 > 3000], 'panel schedule: hub rows')
    assert_bitexact(C2[l2 <= 2500], ref2[l2 <= 2500], 'panel schedule: swept rows are chains already')
    assert (np.abs(C2 - ref2) <= 1e-5 * np.abs(ref2) + 2e-6).all()


@pytest.mark.first_contact
@pytest.mark.parametrize('N', [64, 128, 8, 41])
def test_single_launch_inputs_chain_their_hub_rows(capi, N):
    """Inputs of <= 2^18 nnz / 2^16 rows are ONE launch; with more nnz than the hub threshold that launch is spmm_small_hub: rows
    above the threshold skipped by the row waves and chained by the workgroup afterwards.  Default threshold (16384): rows of
    20000 and 16385 nnz bit for bit the oracle's chain, everything else as before (1e-5; rows <= 64 nnz bit-exact); mean with
    unit weights and the fused epilogue through the same kernel."""
    assert capi.hub_threshold() == 16384
    rng = np.random.default_rng(2)
    M, K = 3000, 30000
    deg = rng.integers(0, 12, M)
    deg[5], deg[2999], deg[77], deg[100] = 20000, 16385, 16384, 700
    rp = np.zeros(M + 1, np.int32)
    rp[1:] = np.cumsum(deg)
    col = np.concatenate([np.sort(rng.choice(K, d, replace=False)) for d in deg]).astype(np.int32)
    val = rng.random(col.size, dtype=np.float32)
    X = rng.random((K, N), dtype=np.float32)
    assert capi.spmm_schedule(capi.SUM, M, K, N, col.size) == 'small'
    ref, _ = oracle.spmm('sum', rp, col, val, X, fma=True, threads=oracle.max_threads())
    hub, short = deg > 16384, deg <= 64
    C, _ = _default(capi, 'sum', rp, col, val, X)
    assert_bitexact(C[hub], ref[hub], 'hub rows of a single-launch input')
    assert_bitexact(C[short], ref[short], 'short rows')
    assert (np.abs(C - ref) <= 1e-5 * np.abs(ref) + 1e-6).all()
    Cm, _ = _default(capi, 'mean', rp, col, None, X)
    refm, _ = oracle.spmm('mean', rp, col, None, X, fma=True, threads=oracle.max_threads())
    assert_bitexact(Cm[hub], refm[hub], 'mean, unit weights')
    d = 'cuda'
    drp, dcol, dval, dX = (torch.from_numpy(a).to(d) for a in (rp, col, val, X))
    bias, rs = torch.linspace(-1, 1, N, device=d), torch.rand(M, device=d) + 0.5
    got, _ = capi.spmm(capi.SUM, drp, dcol, dval, dX, bias=bias, row_scale=rs, relu=True)
    assert torch.equal(got, torch.relu(torch.from_numpy(C).to(d) * rs[:, None] + bias)), 'fused == unfused bit for bit'




@pytest.mark.first_contact
@pytest.mark.parametrize('N', [64, 128, 41])
def test_strict_order_over_the_cached_plan(capi, N):
    """Round 5 (VERDICT r3 #1b): a strict call over a plan takes the plan's strict table (rows > 64 nnz sorted longest first, class
    sizes in its header) - no memset, no classify pass, one launch - and chains exactly what the plan-free strict call chains:
    same bits, which are the oracle's; compact plan and the C entry with provisional counts; the public operator over a
    Storage that has its plan."""
    import dgsparse
    rp, col, st = graphgen.powerlaw_csr(300000, 3000000, alpha=2.0, dmax=30000, seed=23)
    val = graphgen.weights(col.shape[0], 'uniform', 5)
    X = graphgen.features(st['K'], N, 6)
    d = 'cuda'
    drp, dcol, dval, dX = (torch.from_numpy(a).to(d) for a in (rp, col, val, X))
    plan = capi.spmm_plan(drp, dcol, st['K'], N, force=True)
    for alg, fma in ((capi.ALG_STRICT_SUM, True), (capi.ALG_STRICT_NOFMA, False)):
        ref, _ = oracle.spmm('sum', rp, col, val, X, fma=fma, threads=oracle.max_threads())
        C0, _ = capi.spmm(capi.SUM, drp, dcol, dval, dX, algorithm=alg)
        C1, _ = capi.spmm(capi.SUM, drp, dcol, dval, dX, algorithm=alg, plan=plan)
        assert_bitexact(C0.cpu().numpy(), ref, f'strict plan-free fma={fma}')
        assert_bitexact(C1.cpu().numpy(), ref, f'strict over the plan fma={fma}')
    A = dgsparse.SparseTensor(rowptr=drp, col=dcol, values=dval, has_value=True)
    assert A.storage.spmm_plan('csr', N, wait=True)[0] is not None
    out = dgsparse.spmm_sum(A, dX, dgsparse.ALG_STRICT_SUM)
    assert_bitexact(out.cpu().numpy(), oracle.spmm('sum', rp, col, val, X, fma=True, threads=oracle.max_threads())[0],
                    'public operator, strict bits, planned Storage')


@pytest.mark.first_contact
@pytest.mark.parametrize('general,N', [(False, 64), (True, 64), (True, 128), (False, 20)])
def test_few_valued_data_public_operator_strict_and_low_threshold_match_the_reference(capi, monkeypatch, general, N):
    """GPU twin of tests/test_emu_cpu.py::test_few_valued_data_... through the PUBLIC operator: weights and
    features drawn like the reference's fill_random, (float)(rand() % 3) / 10 (example/util/sp_util.hpp:44-48), rows of 2 048 /
    4 096 / 12 288 / 16 384 nnz.  dgsparse.spmm_sum(A, X, ALG_STRICT_NOFMA) is the reference's host loop bit for bit, ALG_STRICT_SUM
    and DGS_HUB_CHAIN=1024 are within 1e-5 of it on every element; the default threshold misses 1e-5 on the 12 288- and 16 384-nnz
    rows - the documented excursion (include/dgsparse_hip.h contract comment), pinned so that it cannot change silently."""
    import dgsparse
    rng = np.random.default_rng(8)
    M, K = (66000, 40000) if general else (48, 40000)
    deg = rng.integers(0, 3, M)
    lens = (2048, 4096, 12288, 16384)
    rows = [5 + 7 * i for i in range(len(lens))]
    for r, L in zip(rows, lens):
        deg[r] = L
    rp = np.zeros(M + 1, np.int32)
    rp[1:] = np.cumsum(deg)
    col = np.concatenate([np.sort(rng.choice(K, d, replace=False)) for d in deg]).astype(np.int32)
    col[-1] = K - 1
    val = (rng.integers(0, 3, col.size) / 10).astype(np.float32)
    X = (rng.integers(0, 3, (K, N)) / 10).astype(np.float32)
    ref = oracle.ref_spmm_sum(rp, col, val, X) if oracle.have_ref() else oracle.spmm('sum', rp, col, val, X, fma=False)[0]
    d = 'cuda'
    A = dgsparse.SparseTensor(rowptr=torch.from_numpy(rp).to(d), col=torch.from_numpy(col).to(d),
                              values=torch.from_numpy(val).to(d), has_value=True)
    Xd = torch.from_numpy(X).to(d)

    def rel(C):
        C = C.cpu().numpy()
        return np.abs(C[rows].astype(np.float64) - ref[rows]) / np.maximum(np.abs(ref[rows]), 1e-30)

    monkeypatch.delenv('DGS_HUB_CHAIN', raising=False)
    assert_bitexact(dgsparse.spmm_sum(A, Xd, dgsparse.ALG_STRICT_NOFMA).cpu().numpy(), ref, 'strict, no contraction == the reference host loop')
    C_strict = dgsparse.spmm_sum(A, Xd, dgsparse.ALG_STRICT_SUM)
    assert rel(C_strict).max() <= 1e-5, rel(C_strict).max(axis=1)
    monkeypatch.setenv('DGS_HUB_CHAIN', '1024')
    A2 = dgsparse.SparseTensor(rowptr=A.storage.rowptr(), col=A.storage.col(), values=A.storage.values(), has_value=True)
    C_low = dgsparse.spmm_sum(A2, Xd, 0)  # (a fresh Storage: the hub hints and the plan are cut at the threshold in force)
    assert torch.equal(C_low[rows], C_strict[rows]), 'rows above the lowered threshold are the strict chains'
    monkeypatch.delenv('DGS_HUB_CHAIN')
    if capi.hub_gate() != 1:
        pytest.skip('hub self-test did not pass on this device: the default is the tree for every row')
    A3 = dgsparse.SparseTensor(rowptr=A.storage.rowptr(), col=A.storage.col(), values=A.storage.values(), has_value=True)
    r = rel(dgsparse.spmm_sum(A3, Xd, 0))
    assert (r[2] > 1e-5).mean() > 0.9 and (r[3] > 1e-5).mean() > 0.9, (r[2].max(), r[3].max())
    assert 1.2e-5 < r[3].max() < 5e-5 and 1.0e-5 < r[2].max() < 4e-5, (r[2].max(), r[3].max())
