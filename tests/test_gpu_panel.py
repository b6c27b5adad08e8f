"""Column-panel SpMM schedule (csrc/spmm_panel.h), forced on small inputs with DGS_PANEL=1 and shrunk panels /
long-row threshold so that every mechanism runs: many panels, several super-blocks, rows that span chunks inside one
panel, rows handed to the unit path, empty rows, unsorted and duplicate columns, K != M.

Bars: max/min values and E bit-exact against the oracle; sum/mean BIT-EXACT against the oracle's sequential fmaf chain
for every row the panel kernel owns (its accumulator is one chain in CSR order), 1e-5 parity for rows that went
through the unit path.
"""
import numpy as np
import pytest
import torch

import oracle
from util import assert_bitexact, assert_sum_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def capi():
    from dgsparse import _capi
    return _capi


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def dense_graph(M, K, deg_lo, deg_hi, seed, sort=True, empty_every=0, hubs=(), dup=False):
    rng = np.random.default_rng(seed)
    deg = rng.integers(deg_lo, deg_hi + 1, size=M)
    if empty_every:
        deg[::empty_every] = 0
    for i, d in hubs:
        deg[i] = d
    deg = np.minimum(deg, K if not dup else deg)
    rowptr = np.zeros(M + 1, dtype=np.int32)
    np.cumsum(deg, out=rowptr[1:])
    col = np.empty(int(rowptr[-1]), dtype=np.int32)
    for i in range(M):
        d = int(deg[i])
        if d == 0:
            continue
        c = rng.integers(0, K, size=d) if dup else rng.choice(K, size=d, replace=False)
        col[rowptr[i]:rowptr[i + 1]] = np.sort(c) if sort else c
    return rowptr, col


def run(capi, reduce, rp, col, val, X):
    C, E = capi.spmm(oracle.REDUCE[reduce], dev(rp), dev(col), None if val is None else dev(val), dev(X))
    torch.cuda.synchronize()
    return C.cpu().numpy(), None if E is None else E.cpu().numpy()


def check(capi, monkeypatch, reduce, rp, col, val, X, tlong, kb=4, lead=1, cross=True):
    monkeypatch.setenv('DGS_PANEL', '1')
    monkeypatch.setenv('DGS_PANEL_KB', str(kb))
    monkeypatch.setenv('DGS_PANEL_LEAD', str(lead))
    monkeypatch.setenv('DGS_PANEL_TLONG', str(tlong))
    assert capi.spmm_schedule(oracle.REDUCE[reduce], rp.size - 1, X.shape[0], X.shape[1], col.size) == 'panel'
    C, E = run(capi, reduce, rp, col, val, X)
    Co, Eo = oracle.spmm(reduce, rp, col, val, X, fma=True)
    lens = np.diff(rp)
    own = lens <= max(tlong, 256)  # rows swept by the panel kernel (the launcher raises tlong to the unit length)
    if reduce in ('max', 'min'):
        assert_bitexact(C, Co, f'{reduce} values')
        assert_bitexact(E, Eo, f'{reduce} E')
    else:
        assert_bitexact(C[own], Co[own], f'{reduce}: rows owned by the panel kernel vs the sequential fmaf chain')
        if (~own).any():
            mean = reduce == 'mean'
            C64 = oracle.spmm_sum_f64(rp, col, val, X, mean=mean)
            S64 = oracle.spmm_sum_f64(rp, col, val, X, mean=mean, absval=True)
            assert_sum_parity(C, Co, C64, S64, 1e-5, 2e-6, f'{reduce}: unit-path rows', lens=lens)
    if not cross:
        return
    # and the forced path must agree with the default dispatch (row-stream on inputs this small)
    monkeypatch.setenv('DGS_PANEL', '0')
    assert capi.spmm_schedule(oracle.REDUCE[reduce], rp.size - 1, X.shape[0], X.shape[1], col.size) == 'rows'
    C0, E0 = run(capi, reduce, rp, col, val, X)
    if reduce in ('max', 'min'):
        assert_bitexact(C, C0, 'panel vs row-stream values')
        assert_bitexact(E, E0, 'panel vs row-stream E')
    else:
        np.testing.assert_allclose(C, C0, rtol=2e-5, atol=1e-5)


@pytest.mark.parametrize('reduce', ['sum', 'mean', 'max', 'min'])
@pytest.mark.parametrize('N', [32, 64, 96, 128, 256, 260, 320, 512])  # > 256: one sweep per 256-feature tile
def test_panel_widths(capi, monkeypatch, reduce, N):
    M, K = 5000, 3000
    rp, col = dense_graph(M, K, 20, 120, seed=N, empty_every=97)
    rng = np.random.default_rng(N + 1)
    val = (rng.random(col.size, dtype=np.float32) - 0.3).astype(np.float32)
    X = (rng.random((K, N), dtype=np.float32) - 0.5).astype(np.float32)
    check(capi, monkeypatch, reduce, rp, col, val, X, tlong=4096, kb=max(4, N // 8))


@pytest.mark.parametrize('reduce', ['sum', 'max', 'min'])
def test_panel_no_values(capi, monkeypatch, reduce):
    rp, col = dense_graph(6000, 3000, 20, 120, seed=5)
    X = (np.random.default_rng(6).random((3000, 64), dtype=np.float32) - 0.5).astype(np.float32)
    check(capi, monkeypatch, reduce, rp, col, None, X, tlong=4096, kb=8)


@pytest.mark.parametrize('reduce', ['sum', 'mean', 'max', 'min'])
def test_panel_long_rows_take_the_unit_path(capi, monkeypatch, reduce):
    M, K = 4200, 6000
    rp, col = dense_graph(M, K, 30, 200, seed=11, hubs=[(0, 5000), (17, 2900), (4199, 1500), (2100, 700)])
    rng = np.random.default_rng(12)
    val = (rng.random(col.size, dtype=np.float32) + 0.25).astype(np.float32)
    X = (rng.random((K, 128), dtype=np.float32) - 0.5).astype(np.float32)
    check(capi, monkeypatch, reduce, rp, col, val, X, tlong=512, kb=64)


@pytest.mark.parametrize('reduce', ['sum', 'max', 'min'])
def test_panel_rows_longer_than_a_chunk_inside_one_panel(capi, monkeypatch, reduce):
    # one panel only (K small), rows of several hundred nnz: the in-visit reload loop runs many times
    M, K = 2048, 900
    rp, col = dense_graph(M, K, 100, 800, seed=21)
    rng = np.random.default_rng(22)
    val = (rng.random(col.size, dtype=np.float32) - 0.5).astype(np.float32)
    X = (rng.random((K, 64), dtype=np.float32) - 0.5).astype(np.float32)
    check(capi, monkeypatch, reduce, rp, col, val, X, tlong=4096, kb=4096)


@pytest.mark.parametrize('reduce', ['sum', 'mean', 'max', 'min'])
def test_panel_unsorted_and_duplicate_columns(capi, monkeypatch, reduce):
    # the cursor consumes a prefix whatever the column order is: unsorted rows only cost cache hits
    M, K = 4500, 2000
    rp, col = dense_graph(M, K, 5, 150, seed=31, sort=False, dup=True, empty_every=50)
    rng = np.random.default_rng(32)
    val = (rng.random(col.size, dtype=np.float32) - 0.5).astype(np.float32)
    X = (rng.integers(-3, 4, size=(K, 32))).astype(np.float32)  # many exact ties for max/min
    check(capi, monkeypatch, reduce, rp, col, val, X, tlong=4096, kb=4)


@pytest.mark.parametrize('lead', [1, 2, 50])
def test_panel_many_superblocks_and_lead(capi, monkeypatch, lead):
    # N = 256 -> 128 rows per workgroup: 100k rows = several super-blocks on 256 CUs
    M, K = 100_000, 1500
    rp, col = dense_graph(M, K, 0, 24, seed=41, dup=True)
    rng = np.random.default_rng(42)
    val = rng.random(col.size, dtype=np.float32)
    X = rng.random((K, 256), dtype=np.float32)
    check(capi, monkeypatch, 'sum', rp, col, val, X, tlong=4096, kb=256, lead=lead)
    check(capi, monkeypatch, 'max', rp, col, val, X, tlong=4096, kb=256, lead=lead)


def test_panel_signed_zero_and_nan_rows(capi, monkeypatch):
    M, K, N = 6000, 700, 32
    rp, col = dense_graph(M, K, 1, 120, seed=51)
    rng = np.random.default_rng(52)
    val = rng.choice(np.array([1.0, -1.0, 0.0, -0.0], dtype=np.float32), size=col.size)
    X = rng.choice(np.array([0.0, -0.0, 1.0, -2.0], dtype=np.float32), size=(K, N))
    X[5, :] = np.nan
    for reduce in ('max', 'min'):
        check(capi, monkeypatch, reduce, rp, col, val, X, tlong=4096, kb=4)


def test_panel_default_dispatch_reddit_like(capi, monkeypatch):
    """Without any override a dense input big enough for the heuristic takes the panel path and matches the oracle."""
    for k in ('DGS_PANEL', 'DGS_PANEL_KB', 'DGS_PANEL_LEAD', 'DGS_PANEL_TLONG'):
        monkeypatch.delenv(k, raising=False)
    M = K = 70_000
    N = 128
    assert capi.spmm_schedule(oracle.REDUCE['sum'], M, K, N, 285 * M) == 'panel'
    assert capi.spmm_schedule(oracle.REDUCE['sum'], M, K, N, 16 * M) == 'rows'  # sparse graph: no panel reuse
    rp, col = dense_graph(M, K, 150, 420, seed=61, hubs=[(3, 20_000), (69_999, 9000)], dup=True)
    rng = np.random.default_rng(62)
    val = rng.random(col.size, dtype=np.float32)
    X = rng.random((K, N), dtype=np.float32)
    C, _ = run(capi, 'sum', rp, col, val, X)
    Co, _ = oracle.spmm('sum', rp, col, val, X, fma=True)
    lens = np.diff(rp)
    own = lens <= 4096
    assert_bitexact(C[own], Co[own], 'panel rows vs sequential chain (default dispatch)')
    C64 = oracle.spmm_sum_f64(rp, col, val, X)
    S64 = oracle.spmm_sum_f64(rp, col, val, X, absval=True)
    assert_sum_parity(C, Co, C64, S64, 1e-5, 2e-6, 'default dispatch', lens=lens)
    Cm, Em = run(capi, 'max', rp, col, val, X)
    Cmo, Emo = oracle.spmm('max', rp, col, val, X, fma=True)
    assert_bitexact(Cm, Cmo, 'max values')
    assert_bitexact(Em, Emo, 'max E')


@pytest.mark.parametrize('N', [32, 128])
def test_panel_masked_sum_for_max_min_backward(capi, monkeypatch, N):
    """max/min backward w.r.t. the dense operand = masked SpMM over the CSC arrays (dgs_spmm_csr_mask_f32): on the
    panel schedule it gathers the saved arg-id row next to the grad row.  Bit-exact against the oracle's sequential
    fmaf chain for the rows the panel kernel owns; hub columns (> tlong entries) go through the unit path."""
    M, K = 6000, 2500
    rp, col = dense_graph(M, K, 30, 110, seed=70 + N)
    rng = np.random.default_rng(71 + N)
    val = (rng.random(col.size, dtype=np.float32) - 0.4).astype(np.float32)
    X = (rng.integers(-3, 4, size=(K, N))).astype(np.float32)  # ties: E must pick the first occurrence
    G = (rng.random((M, N), dtype=np.float32) - 0.5).astype(np.float32)
    colptr, row, tval, _ = oracle.csr2csc(rp, col, val, K)
    _, E = oracle.spmm('max', rp, col, val, X)
    ref = oracle.spmm_mask(colptr, row, tval, G, E, fma=True)
    lens = np.diff(colptr)
    for tlong in (4096, 200):
        monkeypatch.setenv('DGS_PANEL', '1')
        monkeypatch.setenv('DGS_PANEL_KB', '16')
        monkeypatch.setenv('DGS_PANEL_TLONG', str(tlong))
        gX = capi.spmm_mask(dev(colptr), dev(row), dev(tval), dev(G), dev(E)).cpu().numpy()
        own = lens <= max(tlong, 256)
        assert own.any()
        assert_bitexact(gX[own], ref[own], 'masked sum: panel-owned rows vs the sequential chain')
        assert_sum_parity(gX, ref, oracle.spmm_mask_f64(colptr, row, tval, G, E),
                          oracle.spmm_mask_f64(colptr, row, tval, G, E, absval=True), 1e-5, 2e-6, 'masked sum',
                          lens=lens)
        monkeypatch.setenv('DGS_PANEL', '0')
        g0 = capi.spmm_mask(dev(colptr), dev(row), dev(tval), dev(G), dev(E)).cpu().numpy()
        np.testing.assert_allclose(gX, g0, rtol=2e-5, atol=1e-5)


@pytest.mark.parametrize('F', [32, 64, 96, 128, 256, 260, 320, 512])  # > 256: one sweep per 256-feature tile
def test_panel_sddmm(capi, monkeypatch, F):
    """Column-panel SDDMM (csrc/sddmm_panel.h) forced on a small input: D1 rows in LDS, D2 swept in panels, long rows
    cut into segments, every nnz written exactly once; sum, mean and the arg-masked variant against the oracle and
    against the nnz-balanced kernel."""
    M, K = 5000, 2200
    rp, col = dense_graph(M, K, 0, 100, seed=80 + F, hubs=[(1, 2100), (4000, 900), (4999, 1300)], empty_every=61)
    rng = np.random.default_rng(81 + F)
    D1 = (rng.random((M, F), dtype=np.float32) - 0.5).astype(np.float32)
    D2 = (rng.random((K, F), dtype=np.float32) - 0.5).astype(np.float32)
    X = rng.integers(-3, 4, size=(K, F)).astype(np.float32)
    _, E = oracle.spmm('max', rp, col, None, X)
    for tlong in (4096, 300):  # 300: the hub rows become several segments
        res = {}
        for force in ('1', '0'):
            monkeypatch.setenv('DGS_PANEL', force)
            monkeypatch.setenv('DGS_PANEL_KB', '8')
            monkeypatch.setenv('DGS_PANEL_TLONG', str(tlong))
            res[force] = (capi.sddmm(dev(rp), dev(col), dev(D1), dev(D2)).cpu().numpy(),
                          capi.sddmm(dev(rp), dev(col), dev(D1), dev(D2), reduce_op=oracle.REDUCE['mean']).cpu().numpy(),
                          capi.sddmm(dev(rp), dev(col), dev(D1), dev(D2), E=dev(E)).cpu().numpy())
        ref = (oracle.sddmm(rp, col, D1, D2, 'sum', fma=True), oracle.sddmm(rp, col, D1, D2, 'mean', fma=True),
               oracle.sddmm_mask(rp, col, D1, D2, E, fma=True))
        atol = 2e-6 * max(1.0, F / 128)  # dot products of F signed terms around zero: the absolute error grows with F
        for k, what in enumerate(('sum', 'mean', 'masked')):
            np.testing.assert_allclose(res['1'][k], ref[k], rtol=1e-5, atol=atol, err_msg=f'panel sddmm {what}')
            np.testing.assert_allclose(res['1'][k], res['0'][k], rtol=1e-5, atol=atol, err_msg=f'panel vs nnzbal {what}')


def test_panel_sddmm_unsorted_many_superblocks(capi, monkeypatch):
    M, K, F = 150_000, 900, 256
    rp, col = dense_graph(M, K, 0, 12, seed=90, sort=False, dup=True)
    rng = np.random.default_rng(91)
    D1 = rng.random((M, F), dtype=np.float32)
    D2 = rng.random((K, F), dtype=np.float32)
    monkeypatch.setenv('DGS_PANEL', '1')
    monkeypatch.setenv('DGS_PANEL_KB', '128')
    got = capi.sddmm(dev(rp), dev(col), dev(D1), dev(D2)).cpu().numpy()
    np.testing.assert_allclose(got, oracle.sddmm(rp, col, D1, D2, 'sum', fma=True), rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize('reduce', ['sum', 'mean', 'max', 'min'])
def test_panel_autograd_through_the_public_ops(monkeypatch, reduce):
    """dgsparse.spmm_* forward + backward with every kernel on its panel schedule (forward SpMM, SpMM on the CSC arrays
    or masked SpMM, SDDMM or masked SDDMM, HIP gather of the CSC values) against gradients built from the oracle."""
    import dgsparse
    monkeypatch.setenv('DGS_PANEL', '1')
    monkeypatch.setenv('DGS_PANEL_KB', '32')
    monkeypatch.setenv('DGS_PANEL_TLONG', '400')
    M = K = 6000
    N = 64
    rp, col = dense_graph(M, K, 10, 100, seed=100, hubs=[(5, 1500), (5999, 700)])
    rng = np.random.default_rng(101)
    val = (rng.random(col.size, dtype=np.float32) + 0.1).astype(np.float32)
    X = (rng.random((K, N), dtype=np.float32) - 0.5).astype(np.float32)
    G = (rng.random((M, N), dtype=np.float32) - 0.5).astype(np.float32)
    tcsr = torch.sparse_csr_tensor(torch.from_numpy(rp), torch.from_numpy(col), torch.from_numpy(val), size=(M, K),
                                   device='cuda')
    A = dgsparse.SparseTensor.from_torch_sparse_csr_tensor(tcsr.clone().detach(), True, requires_grad=True)
    Xt = dev(X).requires_grad_(True)
    fn = getattr(dgsparse, 'spmm_' + reduce)
    out = fn(A, Xt, 0)
    out.backward(dev(G))
    Co, Eo = oracle.spmm(reduce, rp, col, val, X, fma=True)
    colptr, row, tval, _ = oracle.csr2csc(rp, col, val, K)
    lens = np.diff(rp)
    if reduce in ('max', 'min'):
        assert_bitexact(out.detach().cpu().numpy(), Co, 'forward')
        dX = oracle.spmm_mask(colptr, row, tval, G, Eo, fma=True)
        dA = oracle.sddmm_mask(rp, col, G, X, Eo, fma=True)
    else:
        np.testing.assert_allclose(out.detach().cpu().numpy(), Co, rtol=1e-5, atol=2e-6)
        Gs = G / np.maximum(lens, 1)[:, None].astype(np.float32) if reduce == 'mean' else G
        dX, _ = oracle.spmm('sum', colptr, row, tval, Gs, fma=True)
        dA = oracle.sddmm(rp, col, G, X, reduce, fma=True)
    np.testing.assert_allclose(Xt.grad.cpu().numpy(), dX, rtol=1e-5, atol=4e-6, err_msg='dX')
    np.testing.assert_allclose(A.storage._values.grad.cpu().numpy().ravel(), dA, rtol=1e-5, atol=4e-6, err_msg='dA')


@pytest.mark.parametrize('N', [41, 121])
def test_panel_odd_feature_width_is_padded(capi, monkeypatch, N):
    """A feature width that is not a multiple of 4 cannot use 16-byte lane vectors; when the padded width would take the
    panel schedule the bindings pad the dense operand with zero columns and slice the result.  Feature columns are
    independent chains, so max/min stay bit-exact and sum matches the sequential oracle on panel-owned rows."""
    for k in ('DGS_PANEL', 'DGS_PANEL_KB', 'DGS_PANEL_LEAD', 'DGS_PANEL_TLONG'):
        monkeypatch.delenv(k, raising=False)
    M = K = 40_000 if N > 100 else 100_000   # the dense operand must exceed 16 MB for the panel schedule
    rp, col = dense_graph(M, K, 150, 400, seed=300 + N, dup=True)
    assert capi.spmm_schedule(oracle.REDUCE['sum'], M, K, N, col.size) == 'rows'
    assert capi.spmm_schedule(oracle.REDUCE['sum'], M, K, (N + 3) & ~3, col.size) == 'panel'
    rng = np.random.default_rng(301)
    val = rng.random(col.size, dtype=np.float32)
    X = (rng.random((K, N), dtype=np.float32) - 0.5).astype(np.float32)
    for reduce in ('sum', 'max'):
        C, E = run(capi, reduce, rp, col, val, X)
        Co, Eo = oracle.spmm(reduce, rp, col, val, X, fma=True)
        assert C.shape == (M, N)
        assert_bitexact(C, Co, f'{reduce} (padded to a multiple of 4)')
        if reduce == 'max':
            assert_bitexact(E, Eo, 'E')
    D1 = (rng.random((M, N), dtype=np.float32) - 0.5).astype(np.float32)
    got = capi.sddmm(dev(rp), dev(col), dev(D1), dev(X)).cpu().numpy()
    np.testing.assert_allclose(got, oracle.sddmm(rp, col, D1, X, 'sum', fma=True), rtol=1e-5, atol=2e-6)
    _, Eo = oracle.spmm('max', rp, col, val, X)
    got = capi.sddmm(dev(rp), dev(col), dev(D1), dev(X), E=dev(Eo)).cpu().numpy()
    np.testing.assert_allclose(got, oracle.sddmm_mask(rp, col, D1, X, Eo, fma=True), rtol=1e-5, atol=2e-6)



def test_panel_next_to_a_cu_hogging_stream_and_the_shared_gpu_hint(capi):
    """the column-panel sweep assumes one workgroup per CU, all co-resident.  (1) While a second stream
    keeps the CUs busy the sweep must stay CORRECT (its soft barrier is a bounded hint, never a dependency); (2) with the
    DGS_ALG_SHARED_GPU hint - what dgsparse.dist passes for the product it overlaps with the halo all-to-all - the call
    takes the row-stream schedule instead, gives the same result (sum bit-exact for rows <= 64 nnz, 1e-5 above) and is not
    slower than the contended sweep."""
    import time
    M, K, N = 40000, 40000, 128
    rp, col = dense_graph(M, K, 300, 600, seed=17)  # 'panel' by the default rule (reuse >= 5.5)
    assert capi.spmm_schedule(oracle.SUM, M, K, N, col.size) == 'panel'
    rng = np.random.default_rng(3)
    val = rng.random(col.size, dtype=np.float32)
    X = rng.random((K, N), dtype=np.float32)
    rpd, cold, vald, Xd = dev(rp), dev(col), dev(val), dev(X)
    ref, _ = capi.spmm(oracle.SUM, rpd, cold, vald, Xd)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    A = torch.rand(4096, 4096, device='cuda')

    def contended(algorithm):
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(12):
                A @ A  # noqa: B018  (keeps every CU busy on the side stream for several ms)
        t0 = time.perf_counter()
        out, _ = capi.spmm(oracle.SUM, rpd, cold, vald, Xd, algorithm=algorithm)
        torch.cuda.current_stream().synchronize()
        dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        return out, dt

    out_panel, t_panel = contended(0)
    assert torch.equal(out_panel, ref), 'the sweep under contention must give the very same bits (one chain per row)'
    out_rows, t_rows = contended(capi.ALG_SHARED_GPU)
    lens = np.diff(rp)
    C64 = oracle.spmm_sum_f64(rp, col, val, X)
    assert_sum_parity(out_rows.cpu().numpy(), ref.cpu().numpy(), C64, None, 1e-5, 2e-6, 'row-stream under the hint', lens=lens)
    print(f'contended: panel sweep {t_panel * 1e3:.2f} ms, row-stream (shared-GPU hint) {t_rows * 1e3:.2f} ms')
    assert t_rows < 3.0 * t_panel + 5e-3
