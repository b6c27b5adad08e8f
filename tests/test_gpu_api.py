"""GPU tests of the drop-in Python surface (dgsparse.spmm_* / SparseTensor / torch.ops.dgsparse_spmm.*),
written the way the reference's own tests are (test/test_spmm.py: forward_check / backward_check against
torch.sparse.mm, here via the committed golden vectors generated from exactly that call)."""
import numpy as np
import pytest
import torch

from util import assert_bitexact, assert_close, load_golden

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-5, 2e-6


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
        assert_close(runs[0][0].cpu().numpy(), g['max_dX'], RTOL, ATOL, f'dX deterministic={det}')
        assert_close(runs[0][1].cpu().numpy(), g['max_dA'], RTOL, ATOL, f'dA deterministic={det}')
    assert torch.equal(grads[True][0][0], grads[True][1][0]) and torch.equal(grads[True][0][1], grads[True][1][1])
    assert torch.allclose(grads[True][0][0], grads[False][0][0], rtol=1e-5, atol=2e-6)


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
