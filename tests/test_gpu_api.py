"""GPU tests of the drop-in Python surface (dgsparse.spmm_* / SparseTensor / torch.ops.dgsparse_spmm.*),
written the way the reference's own tests are (test/test_spmm.py: forward_check / backward_check against
torch.sparse.mm, here via the committed golden vectors generated from exactly that call)."""
import os

import numpy as np
import pytest
import torch

from util import assert_bitexact, assert_close, load_golden

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-5, 2e-6


def assert_scatter_close(got, want, g, what):
    """dX of max/min from the single-pass backward: sums of SIGNED terms accumulated with fp32 atomics in whatever order
    the hardware delivers them, so the error is bounded by the condition scale of a column - sum_i |A[i, j]| * max |G| is an
    upper bound of sum |terms| for row j of dX - not by the (possibly cancelled) result.  5 eps of that scale on top of the
    usual bars; the deterministic masked kernels are held to the plain bars."""
    col, val = g['col'], np.abs(g['val']) if 'val' in g else np.ones(g['col'].shape[0], np.float32)
    scale = np.bincount(col, weights=val, minlength=want.shape[0])[:want.shape[0]] * float(np.abs(g['G']).max())
    err = np.abs(np.asarray(got, np.float64) - want)
    bad = err > RTOL * np.abs(want) + ATOL + 3e-7 * scale[:, None]
    assert not bad.any(), f'{what}: {bad.sum()} / {bad.size} outside the scatter bar; worst {err.max()}'


def make(g, requires_grad=True, has_value=True):
    import dgsparse
    M, K = g['rowptr'].shape[0] - 1, int(g['K'])
    tcsr = torch.sparse_csr_tensor(torch.from_numpy(g['rowptr']), torch.from_numpy(g['col']),
                                   torch.from_numpy(g['val']), size=(M, K), device='cuda')
    dcsr = dgsparse.SparseTensor.from_torch_sparse_csr_tensor(tcsr.clone().detach(), has_value,
                                                              requires_grad=requires_grad)
    X = torch.from_numpy(g['X']).cuda().requires_grad_(requires_grad)
    return dcsr, X


GRAD_CASES = ['tiny_nodup_grad_N3', 'tiny_nodup_grad_N32', 'cora_shaped_N32', 'small_weighted_N64',
              'powerlaw4k_signed_grad_N8']


@pytest.mark.parametrize('name', GRAD_CASES)
@pytest.mark.parametrize('reduce', ['sum', 'mean', 'max'])
def test_forward_backward_like_reference_tests(name, reduce):
    import dgsparse
    g = load_golden(name)
    if f'{reduce}_dX' not in g:
        pytest.skip('no golden grads for this reduce')
    dcsr, X = make(g)
    fn = {'sum': dgsparse.spmm_sum, 'mean': dgsparse.spmm_mean, 'max': dgsparse.spmm_max}[reduce]
    out = fn(dcsr, X, 0)
    if reduce == 'max':
        assert_bitexact(out.detach().cpu().numpy(), g['max_out'])
    else:
        assert_close(out.detach().cpu().numpy(), g[f'{reduce}_out'], RTOL, ATOL)
    out.backward(torch.from_numpy(g['G']).cuda())
    if reduce == 'max':
        assert_scatter_close(X.grad.cpu().numpy(), g['max_dX'], g, 'dX (fp32 atomics)')
    else:
        assert_close(X.grad.cpu().numpy(), g[f'{reduce}_dX'], RTOL, ATOL, 'dX')
    assert_close(dcsr.storage._values.grad.cpu().numpy(), g[f'{reduce}_dA'], RTOL, ATOL, 'dA')


def test_max_backward_single_pass_and_deterministic_mode():
    """spmm_max/min backward: by default one pass over the arg ids with fp32 atomics (csrc/arg_backward.hip); under
    torch.use_deterministic_algorithms(True) the two masked kernels, which are bit-reproducible.  Both agree with the
    golden gradients."""
    import dgsparse
    g = load_golden('powerlaw4k_signed_grad_N8')
    G = torch.from_numpy(g['G']).cuda()
    grads = {}
    for det in (False, True):
        torch.use_deterministic_algorithms(det)
        try:
            runs = []
            for _ in range(2):
                dcsr, X = make(g)
                dgsparse.spmm_max(dcsr, X, 0).backward(G)
                runs.append((X.grad.clone(), dcsr.storage._values.grad.clone()))
        finally:
            torch.use_deterministic_algorithms(False)
        grads[det] = runs
        if det:
            assert_close(runs[0][0].cpu().numpy(), g['max_dX'], RTOL, ATOL, 'dX deterministic')
        else:
            assert_scatter_close(runs[0][0].cpu().numpy(), g['max_dX'], g, 'dX single pass (fp32 atomics)')
        assert_close(runs[0][1].cpu().numpy(), g['max_dA'], RTOL, ATOL, f'dA deterministic={det}')
    assert torch.equal(grads[True][0][0], grads[True][1][0]) and torch.equal(grads[True][0][1], grads[True][1][1])
    assert_scatter_close(grads[False][0][0].cpu().numpy(), grads[True][0][0].cpu().numpy().astype(np.float64), g,
                         'single pass vs deterministic')


def test_permuted_values_cache_tracks_in_place_updates():
    """The backward keeps the last `values[csr2csc]`; an in-place update of the edge values (what an optimizer step does)
    must be seen by the next backward, and a different tensor at a recycled address must never alias."""
    import dgsparse
    g = load_golden('small_weighted_N64')
    G = torch.from_numpy(g['G']).cuda() if 'G' in g else None
    dcsr, X = make(g)
    if G is None:
        G = torch.rand(g['rowptr'].shape[0] - 1, X.shape[1], device='cuda')

    def dX():
        X.grad = None
        dgsparse.spmm_sum(dcsr, X, 0).backward(G)
        return X.grad.clone()

    first = dX()
    assert torch.equal(first, dX())  # second call may hit the cache: identical
    with torch.no_grad():
        dcsr.storage._values.mul_(2.0)  # in place: same tensor object, version counter bumped
    assert torch.allclose(dX(), 2.0 * first, rtol=1e-6, atol=0)
    # a fresh values tensor (possibly at the same address) with the same version must not hit
    for scale in (3.0, 5.0):
        d2, _ = make(g)
        with torch.no_grad():
            d2.storage._values.copy_(torch.from_numpy(g['val']).cuda().view_as(d2.storage._values) * scale)
        X.grad = None
        dgsparse.spmm_sum(d2, X, 0).backward(G)
        assert torch.allclose(X.grad, scale * first, rtol=1e-6, atol=1e-6)
        del d2


def test_algorithm_is_a_hint_and_has_value_false():
    import dgsparse
    g = load_golden('cora_shaped_N32')
    dcsr, X = make(g, requires_grad=False)
    ref = dgsparse.spmm_sum(dcsr, X, 0)
    for alg in (1, 2, 3, 7):
        assert torch.equal(dgsparse.spmm_sum(dcsr, X, alg), ref)
    d2, _ = make(g, requires_grad=False, has_value=False)  # weights are ones in this fixture
    assert torch.equal(dgsparse.spmm_sum(d2, X, 0), ref)
    out = torch.ops.dgsparse_spmm.spmm_min(dcsr.storage.rowptr(), dcsr.storage.col(), dcsr.storage.values(),
                                           dcsr.storage.colptr(), dcsr.storage.row(), dcsr.storage.csr2csc(), X,
                                           True, 0)
    assert_bitexact(out.cpu().numpy(), g['min_out'])


def test_storage_csc_and_csr2csc_entry():
    import dgsparse
    g = load_golden('small_weighted_N64')
    dcsr, _ = make(g, requires_grad=False)
    st = dcsr.storage
    assert st.sparse_sizes == (g['rowptr'].shape[0] - 1, int(g['col'].max()) + 1)
    ncol = st.sparse_sizes[1]
    assert_bitexact(st.colptr().cpu().numpy(), g['csc_colptr'][:ncol + 1])
    assert_bitexact(st.row().cpu().numpy(), g['csc_row'])
    assert_bitexact(st.csr2csc().cpu().numpy(), g['csc_perm'])
    colptr, row, vals = dgsparse.csr2csc(dcsr)  # reference test/test_csr2csr.py:40-49
    assert_bitexact(row.cpu().numpy(), g['csc_row'])
    assert_bitexact(vals.cpu().numpy(), g['csc_val'])


def test_sddmm_public_entry():
    import dgsparse
    g = load_golden('cora_shaped_N32')
    dcsr, X = make(g, requires_grad=False)
    out = dgsparse.sddmm(dcsr, torch.from_numpy(g['D1']).cuda(), X)
    assert_close(out.cpu().numpy(), g['ref_sddmm_out'], RTOL, ATOL)


def test_errors_are_loud():
    import dgsparse
    g = load_golden('tiny_N32')
    dcsr, X = make(g, requires_grad=False)
    with pytest.raises(ValueError):
        dgsparse.spmm_sum(dcsr, X[:2], 0)  # fewer dense rows than referenced columns
    with pytest.raises(TypeError):
        dgsparse.spmm_sum(dcsr, X.double(), 0)
    with pytest.raises(RuntimeError):
        dgsparse.spmm_sum(dcsr, X.cpu(), 0)  # no CPU fallback
    with pytest.raises(AssertionError):
        dgsparse.SparseTensor(rowptr=dcsr.storage.rowptr().long(), col=dcsr.storage.col())


def test_dist_engine_single_rank_on_gpu():
    """The partitioned code path (plan, ext relabel, exchange buffer) with one rank on the GPU: must equal the plain
    operator bit for bit.  The N>1 exchange itself is covered by tests/test_dist_cpu.py (gloo, world 2 and 3)."""
    from dgsparse import _capi
    from dgsparse import dist as dd
    sp = dd.synthetic_partition(0, 1, 4096, 12, seed=2, device='cuda')
    X = torch.rand(4096, 64, device='cuda')
    eng = dd.DistSpMM(sp, 64)
    for red, op in (('sum', 0), ('max', 1)):
        C = eng.spmm(X, red)
        Cr, Er = _capi.spmm(op, sp.rowptr, sp.col, sp.val, X)
        assert torch.equal(C, Cr)
        if red == 'max':
            assert torch.equal(eng.last_E, Er)
    # backward w.r.t. B through the engine == transposed SpMM of the plain operator
    G = torch.rand(4096, 64, device='cuda')
    gB = eng.spmm_sum_backward_dense(G)
    colptr, row, tval, _ = _capi.csr2csc(sp.rowptr, sp.col, sp.val, 4096)
    ref, _ = _capi.spmm(0, colptr, row, tval, G)
    assert torch.equal(gB, ref)


def test_standalone_cabi_driver_on_mtx(tmp_path):
    """examples/spmm_mtx (C++ over the C ABI, the counterpart of example/ge-spmm/spmm.cu): builds, loads a
    MatrixMarket file, self-checks all four reduces + SDDMM against its host loop."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(['make', '-s', '-C', os.path.join(root, 'examples')])
    import numpy as np
    rng = np.random.default_rng(0)
    n = 3000
    deg = np.minimum(rng.zipf(1.7, n), 1500)
    rows = np.repeat(np.arange(n), deg)
    cols = rng.integers(0, n, rows.shape[0])
    ent = sorted({(int(max(a, b)), int(min(a, b))) for a, b in zip(rows, cols)})
    p = str(tmp_path / 'g.mtx')
    with open(p, 'w') as f:
        f.write('%%MatrixMarket matrix coordinate pattern symmetric\n')
        f.write(f'{n} {n} {len(ent)}\n')
        f.writelines(f'{a + 1} {b + 1}\n' for a, b in ent)
    out = subprocess.run([os.path.join(root, 'examples', 'spmm_mtx'), p, '48'], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count('passed') == 5 and 'FAILED' not in out.stdout
    assert 'strict order] bit-exact vs the sequential host loop: yes' in out.stdout, out.stdout
    # a larger graph takes the row-stream schedule: the driver then also builds the cached plan through the C ABI
    # (build -> compact -> dgs_spmm_csr_plan_f32) and verifies the four reduces over it
    n = 70000
    rows = np.repeat(np.arange(n), np.minimum(rng.zipf(1.6, n), 6000))
    cols = rng.integers(0, n, rows.shape[0])
    ent = sorted({(int(max(a, b)), int(min(a, b))) for a, b in zip(rows, cols)})
    p2 = str(tmp_path / 'g2.mtx')
    with open(p2, 'w') as f:
        f.write('%%MatrixMarket matrix coordinate pattern symmetric\n')
        f.write(f'{n} {n} {len(ent)}\n')
        f.writelines(f'{a + 1} {b + 1}\n' for a, b in ent)
    out = subprocess.run([os.path.join(root, 'examples', 'spmm_mtx'), p2, '32'], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count('passed') == 5 and out.stdout.count('verification ok') == 4 and 'FAILED' not in out.stdout
    assert 'strict order] bit-exact vs the sequential host loop: yes' in out.stdout, out.stdout  # hub rows of 6000 nnz included


def test_hip_graph_capture_and_replay():
    """The C ABI never allocates or synchronises and launches on the caller's stream, so a whole forward (memset +
    classify + fused + combine, or the single small-input launch) can be captured in a HIP graph and replayed."""
    import dgsparse
    from bench import graphgen
    for name in ('cora', 'arxiv'):  # single-launch path and the 4-launch path
        rp, col, st = graphgen.dataset_shaped(name, seed=0, device='cuda', as_torch=True)
        val = torch.rand(st['nnz'], device='cuda')
        A = dgsparse.SparseTensor(rowptr=rp, col=col, values=val, has_value=True)
        X = torch.rand(st['K'], 64, device='cuda')
        ref = dgsparse.spmm_max(A, X, 0).clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                dgsparse.spmm_max(A, X, 0)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = dgsparse.spmm_max(A, X, 0)
        X.mul_(2.0)  # replay must see the new contents of the captured input buffer
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, dgsparse.spmm_max(A, X, 0))
        assert not torch.equal(out, ref)


def test_hip_graph_capture_panel_schedules(monkeypatch):
    """Same for the column-panel schedules (forced on a mid-size input): memset + classify + panel sweep + unit path,
    and the SDDMM twin (memset of its barrier counter + panel sweep + long-row kernel), captured and replayed."""
    from bench import graphgen
    from dgsparse import _capi
    monkeypatch.setenv('DGS_PANEL', '1')
    monkeypatch.setenv('DGS_PANEL_KB', '64')
    monkeypatch.setenv('DGS_PANEL_TLONG', '512')
    rp, col, st = graphgen.dataset_shaped('arxiv', seed=1, device='cuda', as_torch=True)
    assert _capi.spmm_schedule(_capi.SUM, st['M'], st['K'], 64, st['nnz']) == 'panel'
    val = torch.rand(st['nnz'], device='cuda')
    X = torch.rand(st['K'], 64, device='cuda')
    D1 = torch.rand(st['M'], 64, device='cuda')
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            _capi.spmm(_capi.MAX, rp, col, val, X)
            _capi.sddmm(rp, col, D1, X)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        C, E = _capi.spmm(_capi.MAX, rp, col, val, X)
        W = _capi.sddmm(rp, col, D1, X)
    X.mul_(-1.5)
    g.replay()
    torch.cuda.synchronize()
    C2, E2 = _capi.spmm(_capi.MAX, rp, col, val, X)
    assert torch.equal(C, C2) and torch.equal(E, E2)
    assert torch.equal(W, _capi.sddmm(rp, col, D1, X))


def test_rccl_collectives_single_rank():
    """init_process_group('nccl') + the collective calls of dgsparse/dist.py under torch.distributed.run (1 rank)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    import socket
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
                          '--master-addr', '127.0.0.1', '--master-port', str(port),
                          os.path.join(root, 'tests', 'nccl_selftest.py')], capture_output=True, text=True, env=env,
                         timeout=300)
    assert out.returncode == 0 and 'nccl selftest ok' in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_coo_constructed_sparse_tensor_forward_backward():
    """a caller-supplied COO `row` (CSR order) must not be mistaken for the CSC row indices: forward and both
    gradients of a SparseTensor(row=..., col=...) equal those of the rowptr-constructed one; unsorted COO rows raise."""
    import dgsparse
    g = load_golden('small_weighted_N64')
    rp, col, val = g['rowptr'], g['col'], g['val']
    row = np.repeat(np.arange(rp.shape[0] - 1), np.diff(rp)).astype(np.int32)
    G = torch.from_numpy(g['G']).cuda()
    outs = []
    for kw in (dict(rowptr=torch.from_numpy(rp).cuda()), dict(row=torch.from_numpy(row).cuda()),
               dict(row=torch.from_numpy(row).cuda(), rowptr=torch.from_numpy(rp).cuda())):
        v = torch.from_numpy(val).cuda().requires_grad_()
        A = dgsparse.SparseTensor(col=torch.from_numpy(col).cuda(), values=v, has_value=True, **kw)
        X = torch.from_numpy(g['X']).cuda().requires_grad_()
        out = dgsparse.spmm_sum(A, X, 0)
        out.backward(G)
        outs.append((out.detach(), X.grad.clone(), v.grad.clone()))
        assert_bitexact(A.storage.csc_row().cpu().numpy(), g['csc_row'])
        if 'row' in kw:  # the COO rows stay what the caller gave
            assert torch.equal(A.storage.row().cpu(), torch.from_numpy(row))
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            assert torch.equal(a, b)
    assert_close(outs[1][1].cpu().numpy(), g['sum_dX'], RTOL, ATOL, 'dX of the COO-constructed tensor')
    bad = row.copy()
    bad[[0, -1]] = bad[[-1, 0]]
    with pytest.raises(ValueError):
        dgsparse.SparseTensor(row=torch.from_numpy(bad).cuda(), col=torch.from_numpy(col).cuda())


def test_rectangular_csr2csc_and_square_only_op():
    """dgsparse.csr2csc(SparseTensor) is exact for n_cols > n_rows; the reference-schema op (square-only, like
    the reference's) refuses wide column ids instead of returning garbage."""
    import scipy.sparse as sp
    import dgsparse
    rng = np.random.default_rng(3)
    M, K, nnz = 50, 400, 900
    A = sp.random(M, K, density=nnz / (M * K), format='csr', random_state=3, dtype=np.float32)
    A.sort_indices()
    rp, col, val = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float32)
    st = dgsparse.SparseTensor(rowptr=torch.from_numpy(rp).cuda(), col=torch.from_numpy(col).cuda(),
                               values=torch.from_numpy(val).cuda(), has_value=True)
    colptr, crow, cval = dgsparse.csr2csc(st)
    T = A.tocsc()
    ncol = st.sparse_sizes[1]
    assert_bitexact(colptr.cpu().numpy(), T.indptr[:ncol + 1].astype(np.int32))
    assert_bitexact(crow.cpu().numpy(), T.indices.astype(np.int32))
    assert_bitexact(cval.cpu().numpy(), T.data.astype(np.float32))
    with pytest.raises(RuntimeError, match='square-only'):
        torch.ops.dgsparse_spmm.csr2csc(torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda(),
                                        torch.from_numpy(val).cuda())
    del rng


def test_raw_ops_validate_devices_and_shapes():
    """direct torch.ops calls get cheap argument checks instead of out-of-bounds reads."""
    g = load_golden('tiny_N32')
    rp, col, val = (torch.from_numpy(g[k]).cuda() for k in ('rowptr', 'col', 'val'))
    X = torch.from_numpy(g['X']).cuda()
    with pytest.raises(RuntimeError, match='GPU'):
        torch.ops.dgsparse_spmm.spmm_raw(0, rp.cpu(), col, val, X, True, 0)
    with pytest.raises(RuntimeError, match='one entry per stored element'):
        torch.ops.dgsparse_spmm.spmm_raw(0, rp, col, val[:-1], X, True, 0)
    with pytest.raises(RuntimeError):
        torch.ops.dgsparse_spmm.sddmm(rp, col, X[:, :8].contiguous(), X, 0)
    if torch.cuda.device_count() > 1:
        with pytest.raises(RuntimeError, match='different devices'):
            torch.ops.dgsparse_spmm.spmm_raw(0, rp, col, val, X.to('cuda:1'), True, 0)


@pytest.mark.parametrize('reduce', ['sum', 'mean', 'max'])
def test_public_ops_use_the_storage_plans(reduce):
    """A graph on the row-stream schedule: the Storage builds the forward plan (and, for sum/mean with a dense gradient, the
    plan of the transposed product) on first use, the public operator runs on them, and forward + both gradients still
    meet the bars against the oracle."""
    import dgsparse
    import oracle
    from bench import graphgen
    rp, col, st = graphgen.powerlaw_csr(70000, 900000, alpha=1.9, dmax=20000, seed=9)
    M, K, N = st['M'], st['K'], 16
    val = graphgen.weights(col.shape[0], 'uniform', 1)
    X = graphgen.features(K, N, 2)
    G = graphgen.features(M, N, 3)
    v = torch.from_numpy(val).cuda().requires_grad_()
    A = dgsparse.SparseTensor(rowptr=torch.from_numpy(rp).cuda(), col=torch.from_numpy(col).cuda(), values=v,
                              has_value=True)
    Xd = torch.from_numpy(X).cuda().requires_grad_()
    fn = {'sum': dgsparse.spmm_sum, 'mean': dgsparse.spmm_mean, 'max': dgsparse.spmm_max}[reduce]
    # plans are built lazily (from the 4th use on, on a side stream): have them now, so that this call runs on them
    assert A.storage.spmm_plan('csr', N, wait=True)[0] is not None and A.storage._plans['csr'].ready is not None
    if reduce != 'max':
        assert A.storage.spmm_plan('csc', N, wait=True)[0] is not None
    out = fn(A, Xd, 0)
    out.backward(torch.from_numpy(G).cuda())
    if reduce != 'max':
        assert A.storage._tvalues is not None
    Co, Eo = oracle.spmm(reduce, rp, col, val, X, fma=True)
    lens = np.diff(rp)
    if reduce == 'max':
        assert_bitexact(out.detach().cpu().numpy(), Co)
    else:
        assert_close(out.detach().cpu().numpy(), Co, 2e-5, ATOL, f'{reduce} forward through the plan')
    # gradients against the formulas on the oracle (A^T G with per-row scaling for mean; arg-masked for max)
    colptr, crow, tval, _ = oracle.csr2csc(rp, col, val, K)
    if reduce == 'sum':
        gX, _ = oracle.spmm('sum', colptr, crow, tval, G, fma=True)
        gW = oracle.sddmm(rp, col, G, X, fma=True)
    elif reduce == 'mean':
        Gs = (G / np.maximum(lens, 1)[:, None]).astype(np.float32)
        gX, _ = oracle.spmm('sum', colptr, crow, tval, Gs, fma=True)
        gW = oracle.sddmm(rp, col, Gs, X, fma=True)
    else:
        row_of = np.repeat(np.arange(M), lens)
        hit = Eo[row_of] == col[:, None]  # [nnz, N]: this entry is the arg of (row, f)
        gW = (hit * G[row_of] * X[col]).sum(1).astype(np.float32)
        gX = np.zeros((K, N), np.float64)
        np.add.at(gX, col, hit * (val[:, None].astype(np.float64) * G[row_of]))
    assert_close(Xd.grad.cpu().numpy(), gX, 3e-5, 1e-5, f'{reduce} dX through the transposed plan')
    assert_close(v.grad.cpu().numpy(), gW, 3e-5, 1e-5, f'{reduce} dA')


def test_plan_lifecycle_lazy_async_shared(monkeypatch):
    """The reference's operator has no per-matrix setup (dgsparse/spmm.py:5-28): a matrix used a few times must never pay
    for a plan, the build must not synchronise the host, and Storages over the same arrays share one plan."""
    import dgsparse
    from bench import graphgen
    from dgsparse import storage as dst
    monkeypatch.delenv('DGS_PLAN_AFTER', raising=False)
    rp, col, st = graphgen.powerlaw_csr(70000, 900000, alpha=1.9, dmax=20000, seed=10, device='cuda', as_torch=True)
    N = 64
    X = torch.rand((st['K'], N), device='cuda')
    A = dgsparse.SparseTensor(rowptr=rp, col=col, values=None, has_value=False)
    ref = dgsparse.spmm_sum(A, X, 0)
    for k in range(dst._plan_after() - 1):
        dgsparse.spmm_sum(A, X, 0)
    sp = A.storage._plans['csr']
    assert sp.calls == dst._plan_after() and sp.prov is None and sp.ready is None, 'no build in the first uses'
    out = dgsparse.spmm_sum(A, X, 0)  # this use queues the build on the caller's stream and already runs on its tables,
    assert sp.prov is not None or sp.ready is not None  # with provisional (upper-bound) counts: no host synchronisation
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-6)
    prov_buf = sp.prov[0] if sp.prov is not None else None
    torch.cuda.synchronize()
    out = dgsparse.spmm_sum(A, X, 0)  # the build's event has completed: this call swaps in the compact plan
    assert sp.ready is not None and sp.prov is None
    assert prov_buf is None or sp.ready[0].numel() < prov_buf.numel() // 4, 'the worst-case build buffer is dropped'
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-6)
    for red in (dgsparse.spmm_max, dgsparse.spmm_mean):  # provisional counts through the other reduces as well
        rp3, col3 = rp.clone(), col.clone()
        D = dgsparse.SparseTensor(rowptr=rp3, col=col3, values=None, has_value=False)
        want = red(D, X, 0)
        D.storage._plans['csr'].calls = dst._plan_after()
        got = red(D, X, 0)
        assert D.storage._plans['csr'].prov is not None or D.storage._plans['csr'].ready is not None
        assert torch.equal(got, want) if red is dgsparse.spmm_max else torch.allclose(got, want, rtol=1e-5, atol=1e-6)
    # a second SparseTensor over the same index arrays shares the plan object (no second build)
    B = dgsparse.SparseTensor(rowptr=rp, col=col, values=torch.rand(st['nnz'], device='cuda'), has_value=True)
    assert B.storage.spmm_plan('csr', N)[0] is sp.ready[0] and B.storage._plans['csr'] is sp
    # the decision is per feature width: a width on another schedule gets no plan, and does not poison the others
    assert A.storage.spmm_plan('csr', 64)[0] is not None
    # a fresh matrix inside a stream capture: nothing is built or polled, the capture survives, replay is right
    rp2, col2 = rp.clone(), col.clone()
    C = dgsparse.SparseTensor(rowptr=rp2, col=col2, values=None, has_value=False)
    monkeypatch.setenv('DGS_PLAN_AFTER', '0')
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        dgsparse.spmm_sum(A, X, 0)  # warm-up outside the capture (allocator)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            y = dgsparse.spmm_sum(C, X, 0)
    assert C.storage._plans['csr'].prov is None and C.storage._plans['csr'].ready is None
    g.replay()
    torch.cuda.synchronize()
    assert torch.allclose(y, ref, rtol=1e-5, atol=1e-6)


@pytest.mark.first_contact
def test_second_tensor_over_the_same_arrays_while_the_plan_is_provisional(monkeypatch):
    """a new SparseTensor per layer / iteration over the same graph lands inside the ~1.5 ms window in which
    the shared plan is still provisional; it must get the build buffer (same stream) like the first one, never a KeyError."""
    import dgsparse
    from bench import graphgen
    from dgsparse import storage as dst
    monkeypatch.delenv('DGS_PLAN_AFTER', raising=False)
    rp, col, st = graphgen.powerlaw_csr(70000, 900000, alpha=1.9, dmax=20000, seed=14, device='cuda', as_torch=True)
    N = 64
    X = torch.rand((st['K'], N), device='cuda')
    A = dgsparse.SparseTensor(rowptr=rp, col=col, values=None, has_value=False)
    ref = dgsparse.spmm_sum(A, X, 0)
    torch.cuda.synchronize()
    for k in range(dst._plan_after()):
        dgsparse.spmm_sum(A, X, 0)  # the last of these queues the build: no synchronisation from here on
    sp = A.storage._plans['csr']
    outs = []
    for k in range(4):
        B = dgsparse.SparseTensor(rowptr=rp, col=col, values=None, has_value=False)
        outs.append(dgsparse.spmm_sum(B, X, 0))
        assert B.storage._plans['csr'] is sp
    torch.cuda.synchronize()
    for o in outs:
        assert torch.allclose(o, ref, rtol=1e-5, atol=1e-6)
    # and a sharer on ANOTHER stream while provisional: plan-free there (None), not an error
    C = dgsparse.SparseTensor(rowptr=rp.clone(), col=col.clone(), values=None, has_value=False)
    for k in range(dst._plan_after() + 1):
        dgsparse.spmm_sum(C, X, 0)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        D = dgsparse.SparseTensor(rowptr=C.storage.rowptr(), col=C.storage.col(), values=None, has_value=False)
        out = dgsparse.spmm_sum(D, X, 0)
    torch.cuda.synchronize()
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-6)


@pytest.mark.first_contact
def test_schedule_switch_is_a_function_of_the_use_count_and_reproducible_mode(monkeypatch):
    """the plan-free and the planned schedule fold rows of 65 .. 8192 nnz with different trees, so WHEN a matrix
    switches must not depend on timing.  (i) Two fresh tensors over clones of the same arrays, one driven in a tight loop and
    one with a device synchronisation after every call, agree bit for bit call by call; calls 1 .. DGS_PLAN_AFTER equal each
    other and so do all later ones.  (ii) DGS_REPRODUCIBLE=1 / torch.use_deterministic_algorithms: one schedule for the whole
    life of the matrix - plan-free until spmm_plan(wait=True), planned after it."""
    import dgsparse
    from bench import graphgen
    from dgsparse import storage as dst
    monkeypatch.delenv('DGS_PLAN_AFTER', raising=False)
    monkeypatch.delenv('DGS_REPRODUCIBLE', raising=False)
    rp, col, st = graphgen.powerlaw_csr(70000, 900000, alpha=1.9, dmax=20000, seed=12, device='cuda', as_torch=True)
    N = 64
    g = torch.Generator(device='cuda')
    g.manual_seed(5)
    X = torch.rand((st['K'], N), generator=g, device='cuda')
    val = torch.rand(st['nnz'], generator=g, device='cuda')
    n_calls, after = 8, dst._plan_after()
    runs = []
    for sync in (False, True):
        A = dgsparse.SparseTensor(rowptr=rp.clone(), col=col.clone(), values=val.clone(), has_value=True)
        outs = []
        for _ in range(n_calls):
            outs.append(dgsparse.spmm_sum(A, X, 0))
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        runs.append(outs)
        assert A.storage._plans['csr'].prov is not None or A.storage._plans['csr'].ready is not None, 'the plan was started'
    for k in range(n_calls):
        assert torch.equal(runs[0][k], runs[1][k]), f'call {k + 1}: tight loop and synchronised loop differ'
    for k in range(1, after):
        assert torch.equal(runs[0][k], runs[0][0]), f'call {k + 1} is plan-free like call 1'
    for k in range(after + 1, n_calls):
        assert torch.equal(runs[0][k], runs[0][after]), f'call {k + 1} is planned like call {after + 1}'
    # (ii) reproducible mode
    monkeypatch.setenv('DGS_REPRODUCIBLE', '1')
    A = dgsparse.SparseTensor(rowptr=rp.clone(), col=col.clone(), values=val.clone(), has_value=True)
    outs = [dgsparse.spmm_sum(A, X, 0) for _ in range(n_calls)]
    sp = A.storage._plans['csr']
    assert sp.prov is None and sp.ready is None, 'nothing is built behind the caller in reproducible mode'
    assert all(torch.equal(o, outs[0]) for o in outs) and torch.equal(outs[0], runs[0][0])
    assert A.storage.spmm_plan('csr', N, wait=True)[0] is not None
    outs = [dgsparse.spmm_sum(A, X, 0) for _ in range(4)]
    assert all(torch.equal(o, outs[0]) for o in outs) and torch.equal(outs[0], runs[0][after])
    monkeypatch.delenv('DGS_REPRODUCIBLE')
    torch.use_deterministic_algorithms(True)
    try:
        assert dst._reproducible()
    finally:
        torch.use_deterministic_algorithms(False)


@pytest.mark.first_contact
def test_two_host_threads_two_streams_two_plans():
    """the launchers keep no mutable global state (the tuning snapshot is immutable, per-device facts are
    atomics), so two host threads may launch concurrently, each on its own stream with its own plan and workspace: every
    result of every iteration must be the single-threaded one, bit for bit (plan-free / planned / max over the shared plan /
    the dense-graph schedule whose launcher sets a per-instantiation attribute on first use)."""
    import threading
    from bench import graphgen
    from dgsparse import _capi
    graphs = []
    for seed in (31, 32):
        rp, col, st = graphgen.powerlaw_csr(60000, 800000, alpha=1.9, dmax=15000, seed=seed, device='cuda', as_torch=True)
        g = torch.Generator(device='cuda')
        g.manual_seed(seed)
        val = torch.rand(st['nnz'], generator=g, device='cuda')
        X = torch.rand((st['K'], 64), generator=g, device='cuda')
        plan = _capi.spmm_plan(rp, col, st['K'], 64, force=True)
        graphs.append((rp, col, val, X, plan))
    torch.cuda.synchronize()

    def work(gr, out):
        rp, col, val, X, plan = gr
        out.append(_capi.spmm(_capi.SUM, rp, col, val, X)[0])
        out.append(_capi.spmm(_capi.SUM, rp, col, val, X, plan=plan)[0])
        out.extend(_capi.spmm(_capi.MAX, rp, col, val, X, plan=plan))
        out.append(_capi.spmm(_capi.MEAN, rp, col, val, X, plan=plan)[0])

    want = []
    for gr in graphs:
        o = []
        work(gr, o)
        want.append(o)
    torch.cuda.synchronize()
    errors = []

    def thread(i):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(20):
                    o = []
                    work(graphs[i], o)
                    s.synchronize()
                    for a, b in zip(o, want[i]):
                        if not torch.equal(a, b):
                            errors.append(f'thread {i}: result differs from the single-threaded run')
                            return
        except Exception as e:  # noqa: BLE001
            errors.append(f'thread {i}: {e!r}')

    ts = [threading.Thread(target=thread, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_gin_cached_neighbourhood_is_keyed_on_the_graph():
    """cached=True must not reuse the first adjacency for another edge_index."""
    from dgsparse import nn as dnn
    conv = dnn.GINConv(None, 'sum', cached=True).cuda()
    e1 = torch.tensor([[0, 1, 2], [1, 2, 0]], device='cuda')
    e2 = torch.tensor([[0, 1, 2, 3], [1, 0, 3, 2]], device='cuda')
    X = torch.eye(4, device='cuda')
    a = conv(e1, X[:3, :3].contiguous(), 3)
    d1 = conv._cached_dcsr
    assert conv(e1, X[:3, :3].contiguous(), 3).equal(a) and conv._cached_dcsr is d1
    b = conv(e2, X, 4)
    assert conv._cached_dcsr is not d1 and b.shape == (4, 4)
    assert torch.equal(b, X + X[[1, 0, 3, 2]])


@pytest.mark.first_contact
def test_hub_self_test_passes_on_this_device_and_gates_the_default(monkeypatch):
    """The default sum / mean chain their hub rows only on a device where the library's self-test has compared that chain, bit
    for bit, with a one-thread-per-element sequential kernel (include/dgsparse_hip.h "Device gate").  On a healthy
    MI355X it passes - so the default IS the chained schedule - and an explicit DGS_HUB_CHAIN wins both ways."""
    from dgsparse import _capi
    monkeypatch.delenv('DGS_HUB_CHAIN', raising=False)
    _capi.ensure_hub_selftest(torch.device('cuda', torch.cuda.current_device()))
    nsh = int(_capi._lib.dgs_spmm_selftest_hub_shapes())
    assert nsh == 14
    assert _capi.hub_gate() == 1, f'the hub-chain self-test FAILED on this device (per shape: {_capi.selftest_detail()[16:16 + nsh]})'
    assert _capi.selftest_detail()[16:16 + nsh] == [0] * nsh, 'every hub shape (hub-workgroup family x schedule) bit-exact'
    assert _capi.hub_threshold() == 16384
    monkeypatch.setenv('DGS_HUB_CHAIN', '0')
    assert _capi.hub_threshold() == 0
    monkeypatch.setenv('DGS_HUB_CHAIN', '3000')
    assert _capi.hub_threshold() == 3000
    monkeypatch.delenv('DGS_HUB_CHAIN')
    assert _capi.hub_threshold() == 16384
    # the C entry again, by hand: idempotent, reports 1
    nb = int(_capi._lib.dgs_spmm_hub_selftest_bytes())
    scratch = torch.empty(nb, dtype=torch.uint8, device='cuda')
    assert _capi._lib.dgs_spmm_hub_selftest(scratch.data_ptr(), nb, torch.cuda.current_stream().cuda_stream) == 1
    assert _capi._lib.dgs_spmm_hub_selftest(scratch.data_ptr(), nb - 1, torch.cuda.current_stream().cuda_stream) == -2
    # a process that pins the threshold does not pay for the test (no launch, no sync: returns 1 at once, the gate is not consulted)
    monkeypatch.setenv('DGS_HUB_CHAIN', '3000')
    _capi.reload_tuning()
    assert _capi._lib.dgs_spmm_hub_selftest(None, 0, torch.cuda.current_stream().cuda_stream) == 1
    monkeypatch.delenv('DGS_HUB_CHAIN')
    _capi.reload_tuning()


@pytest.mark.first_contact
def test_gate_verdict_cache_spares_the_second_process_the_self_test(monkeypatch, tmp_path):
    """DGS_GATE_CACHE: a PASS of the hub self-test is kept per (library binary, device model, runtime); a later process - here: the
    same one with its per-process state wiped - adopts it through dgs_spmm_hub_gate_assume instead of running the test again."""
    from dgsparse import _capi
    monkeypatch.delenv('DGS_HUB_CHAIN', raising=False)
    monkeypatch.setenv('DGS_GATE_CACHE', str(tmp_path))
    dev = torch.device('cuda', torch.cuda.current_device())
    _capi._selftested.discard(dev.index)
    _capi._lib.dgs_spmm_hub_gate_assume(0)
    _capi.ensure_hub_selftest(dev)                      # runs the test, writes the verdict
    assert _capi.hub_gate() == 1
    files = [f for f in os.listdir(tmp_path) if f.startswith('dgs_gate_')]
    assert len(files) == 1
    _capi._selftested.discard(dev.index)
    _capi._lib.dgs_spmm_hub_gate_assume(0)
    assert _capi.hub_gate() == 0

    def boom(*a):
        raise AssertionError('the self-test ran although a cached verdict exists')
    monkeypatch.setattr(_capi._lib, 'dgs_spmm_hub_selftest', boom)
    _capi.ensure_hub_selftest(dev)                      # adopts it
    assert _capi.hub_gate() == 1 and _capi.hub_threshold() == 16384


@pytest.mark.first_contact
def test_fold_self_test_passes_for_every_family_of_partial_row(monkeypatch):
    """the in-kernel fold's hand-over (sc1 stores -> drain -> agent-scope counter -> sc1 loads) checked by the device
    for EVERY family of partial row the launchers can pick - whole-line slots (N = 64, 32), slots sharing a 128-byte line two / four
    / eight to a line (N = 16, 8, 4), scalar-lane slots written with 4-byte atomics (N = 20, 3), two feature tiles with their own
    arrival counters (N = 256), 512-byte slots (N = 128) - sum / max / min, rows of 2 .. 59 units, three rounds each, the last one with a streaming kernel
    loading the fabric from a second stream; then ten more loaded rounds of the line-sharing and scalar families.  The fold is
    opt-in (DGS_FOLD=1 | 2): the gate only matters to DGS_FOLD=2, and this test is what says whether opting in is safe here."""
    from dgsparse import _capi
    monkeypatch.delenv('DGS_FOLD', raising=False)
    _capi.reload_tuning()
    assert _capi._lib.dgs_spmm_selftest_families() == 9
    verdict, fam = _capi.fold_selftest(rounds=3, load=True)
    assert verdict == 1 and fam == [0] * 9, f'in-kernel fold self-test FAILED: mismatches per family {fam}'
    assert _capi.fold_gate() == 1
    verdict, fam = _capi.fold_selftest(rounds=10, load=True, families=[2, 3, 4, 5, 8])
    assert verdict == 1 and fam == [0] * 9, f'in-kernel fold self-test FAILED under repetition: mismatches per family {fam}'
    # the default does not fold in the kernel whatever the gate says; DGS_FOLD=2 does where the gate is up
    torch.cuda.synchronize()


@pytest.mark.first_contact
def test_storage_tells_the_launches_its_longest_row_and_column(monkeypatch):
    """single-launch inputs are routed by the longest ROW, not by nnz.  The Storage learns the longest row in the
    one sync its construction already has and the longest column without another one; spmm_sum / spmm_mean pass the hints on
    (forward: rows, backward: columns).  Same bits with and without them; a matrix WITH a hub row gets no hint."""
    import dgsparse
    import oracle
    from dgsparse import _capi
    monkeypatch.delenv('DGS_HUB_CHAIN', raising=False)
    rng = np.random.default_rng(9)
    M = K = 6000
    deg = rng.integers(0, 10, M)
    deg[17], deg[4000] = 280, 78
    rp = np.zeros(M + 1, np.int32)
    rp[1:] = np.cumsum(deg)
    col = np.concatenate([np.sort(rng.choice(K, d, replace=False)) for d in deg]).astype(np.int32)
    col[-1] = K - 1
    val = rng.random(col.size, dtype=np.float32)
    d = 'cuda'
    drp, dcol, dval = (torch.from_numpy(a).to(d) for a in (rp, col, val))
    A = dgsparse.SparseTensor(rowptr=drp, col=dcol, values=dval, has_value=True)
    st = A.storage
    assert st._max_row_len == 280
    torch.cuda.synchronize()
    want = _capi.ALG_NO_HUB_ROWS | _capi.ALG_NO_HUB_COLS
    assert st.hub_hints() == want and st._hints == (16384, want)
    assert st._max_col_len == int(np.bincount(col, minlength=K).max())
    X = torch.rand((K, 64), device=d, requires_grad=True)
    out = dgsparse.spmm_sum(A, X, 0)
    plain, _ = _capi.spmm(_capi.SUM, drp, dcol, dval, X.detach())  # no hint: the single-launch hub kernel (nnz > threshold)
    assert torch.equal(out.detach(), plain)
    G = torch.rand_like(out)
    out.backward(G)
    colptr, row, tval, _ = oracle.csr2csc(rp, col, val, K)
    gX, _ = oracle.spmm('sum', colptr, row, tval, G.cpu().numpy(), fma=True)
    assert_close(X.grad.cpu().numpy(), gX, 1e-5, 2e-6, 'dX with the column hint')
    monkeypatch.setenv('DGS_HUB_CHAIN', '200')  # now row 17 IS a hub row: no row hint, the column hint survives if true
    bits = st.hub_hints()
    assert not bits & _capi.ALG_NO_HUB_ROWS
    out2 = dgsparse.spmm_sum(A, X.detach(), 0)
    ref, _ = oracle.spmm('sum', rp, col, val, X.detach().cpu().numpy(), fma=True)
    assert np.array_equal(out2.cpu().numpy()[17].view(np.int32), ref[17].view(np.int32)), 'the hub row is chained'
