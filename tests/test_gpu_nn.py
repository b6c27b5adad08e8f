"""GCN / GIN layers: forward + backward through SpMM + SDDMM + csr2csc against a dense torch
computation of the same model on a small graph."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def graph(n=300, e=2500, seed=0):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, e), generator=g)
    ei = torch.unique(ei, dim=1)  # simple graph
    return ei.cuda(), n


def dense_adj(ei, n, self_loops):
    A = torch.zeros(n, n, device='cuda')
    A[ei[0], ei[1]] = 1.0
    if self_loops:
        A.fill_diagonal_(1.0)
    return A


def test_gcn_forward_backward_matches_dense():
    from dgsparse import nn as dnn
    ei, n = graph()
    torch.manual_seed(0)
    model = dnn.GCN(32, 7, 16).cuda()
    x = torch.rand(n, 32, device='cuda', requires_grad=True)
    dcsr = dnn.get_gcn_dcsr_from_edge_index(ei, n)
    out = model(dcsr, x)
    out.square().sum().backward()
    gx, gw = x.grad.clone(), model.conv1.W.weight.grad.clone()
    gA = dcsr.storage._values.grad.clone()
    # dense reference
    A = dense_adj(ei, n, True)
    dis = A.sum(1).pow(-0.5)
    Ah = (dis[:, None] * A * dis[None, :]).requires_grad_()
    x2 = x.detach().clone().requires_grad_()
    model.zero_grad()
    ref = Ah @ model.conv2.W(torch.relu(Ah @ model.conv1.W(x2)))
    ref.square().sum().backward()
    assert torch.allclose(out, ref, rtol=1e-4, atol=1e-5)
    assert torch.allclose(gx, x2.grad, rtol=1e-4, atol=1e-5)
    assert torch.allclose(gw, model.conv1.W.weight.grad, rtol=1e-4, atol=1e-5)
    st = dcsr.storage
    rows = torch.repeat_interleave(torch.arange(n, device='cuda'), (st.rowptr()[1:] - st.rowptr()[:-1]).long())
    assert torch.allclose(gA, Ah.grad[rows, st.col().long()], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('agg', ['sum', 'max', 'mean'])
def test_gin_matches_dense(agg):
    from dgsparse import nn as dnn
    ei, n = graph(seed=1)
    torch.manual_seed(1)
    model = dnn.GIN(16, 5, 24, aggregator_type=agg, init_eps=0.1, learn_eps=True, cached=True).cuda()
    x = torch.rand(n, 16, device='cuda', requires_grad=True)
    out = model(ei, x, n)
    out.sum().backward()
    A = dense_adj(ei, n, False)

    def aggr(h):
        if agg == 'sum':
            return A @ h
        if agg == 'mean':
            return (A @ h) / A.sum(1).clamp(min=1)[:, None]
        m = torch.where(A[:, :, None] > 0, h[None, :, :], torch.full_like(h[None, :, :], -3e38)).max(1).values
        return torch.where(A.sum(1)[:, None] > 0, m, torch.zeros_like(m))

    x2 = x.detach().clone().requires_grad_()
    h = torch.relu(model.conv1.apply_func((1 + model.conv1.eps) * x2 + aggr(x2)))
    ref = torch.relu(model.conv2.apply_func((1 + model.conv2.eps) * h + aggr(h)))
    assert torch.allclose(out, ref, rtol=1e-4, atol=1e-5)
    gx = x.grad.clone()
    ref.sum().backward()
    assert torch.allclose(gx, x2.grad, rtol=1e-4, atol=1e-4)
    assert model.conv1._cached_dcsr is not None


def test_gcn_training_example_runs():
    """examples/train_gcn.py: 2-layer GCN trained end to end (SpMM fwd, SDDMM + transposed SpMM bwd); the loss
    trajectory must match the same model on torch.sparse.mm."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'examples', 'train_gcn.py'), '--dataset', 'cora', '--epochs',
                          '10'], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'loss trajectories match' in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]


@pytest.mark.parametrize('shape', ['small', 'rows', 'rows+plan', 'panel'])
@pytest.mark.parametrize('N', [16, 64, 41])
def test_fused_epilogue_equals_the_unfused_ops_bit_for_bit(shape, N, monkeypatch):
    """relu(row_scale * (A @ X) + bias) inside the SpMM's row-end store (dgs_spmm_csr_ex_f32) == the three separate torch
    ops on the plain SpMM result, bit for bit, on every schedule; mean too; max rejects an epilogue."""
    from bench import graphgen
    from dgsparse import _capi
    if shape == 'small':
        rp, col, st = graphgen.powerlaw_csr(3000, 40000, alpha=1.9, dmax=1500, seed=3)
    elif shape == 'panel':
        rp, col, st = graphgen.powerlaw_csr(9000, 1_500_000, alpha=2.6, dmax=6000, seed=5)
        monkeypatch.setenv('DGS_PANEL', '1')
        monkeypatch.setenv('DGS_PANEL_TLONG', '2500')
    else:
        rp, col, st = graphgen.powerlaw_csr(70000, 900000, alpha=1.9, dmax=20000, seed=9)
    deg = np.diff(rp)
    keep = np.ones(col.shape[0], bool)
    for r in range(5, st['M'], 97):  # some empty rows: their output is the epilogue of 0
        keep[rp[r]:rp[r + 1]] = False
        deg[r] = 0
    col = np.ascontiguousarray(col[keep])
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    assert (np.diff(rp) == 0).any() and np.diff(rp).max() > 64
    M, K = st['M'], st['K']
    d = 'cuda'
    rpt, colt = torch.from_numpy(rp).to(d), torch.from_numpy(col).to(d)
    val = torch.rand(col.shape[0], device=d) - 0.3
    X = torch.rand((K, N), device=d) - 0.5
    bias = torch.rand(N, device=d) - 0.5
    rs = torch.rand(M, device=d) + 0.5
    plan = _capi.spmm_plan(rpt, colt, K, N, force=True) if shape == 'rows+plan' else None
    for op in (_capi.SUM, _capi.MEAN):
        base, _ = _capi.spmm(op, rpt, colt, val, X, plan=plan)
        for kw in (dict(relu=True), dict(bias=bias), dict(row_scale=rs), dict(bias=bias, row_scale=rs, relu=True)):
            got, _ = _capi.spmm(op, rpt, colt, val, X, plan=plan, **kw)
            want = base
            if 'row_scale' in kw:
                want = want * rs[:, None]
            if 'bias' in kw:
                want = want + bias
            if kw.get('relu'):
                want = torch.relu(want)
            assert torch.equal(got, want), (shape, N, op, list(kw))
    with pytest.raises(ValueError):
        _capi.spmm(_capi.MAX, rpt, colt, val, X, relu=True)


def test_fused_spmm_autograd_and_gcn_layer():
    """spmm_sum_fused: forward == relu(spmm_sum + bias) exactly, gradients w.r.t. X, the edge values and the bias equal
    those of the unfused graph; GCN (conv1 with the fused ReLU) equals the same model on the unfused operators."""
    import dgsparse
    from bench import graphgen
    from dgsparse import nn as dnn
    rp, col, st = graphgen.powerlaw_csr(20000, 300000, alpha=2.0, dmax=3000, seed=4)
    M, N = st['M'], 32
    d = 'cuda'
    v1 = (torch.rand(col.shape[0], device=d) - 0.2).requires_grad_()
    v2 = v1.detach().clone().requires_grad_()
    A1 = dgsparse.SparseTensor(rowptr=torch.from_numpy(rp).to(d), col=torch.from_numpy(col).to(d), values=v1, has_value=True)
    A2 = dgsparse.SparseTensor(rowptr=A1.storage.rowptr(), col=A1.storage.col(), values=v2, has_value=True)
    X1 = (torch.rand((M, N), device=d) - 0.5).requires_grad_()
    X2 = X1.detach().clone().requires_grad_()
    b1 = (torch.rand(N, device=d) - 0.5).requires_grad_()
    b2 = b1.detach().clone().requires_grad_()
    G = torch.rand((M, N), device=d)
    y1 = dnn.spmm_sum_fused(A1, X1, bias=b1, relu=True)
    y2 = torch.relu(dgsparse.spmm_sum(A2, X2, 0) + b2)
    assert torch.equal(y1, y2)
    y1.backward(G)
    y2.backward(G)
    assert torch.allclose(X1.grad, X2.grad, rtol=1e-5, atol=1e-6)
    assert torch.allclose(v1.grad, v2.grad, rtol=1e-5, atol=1e-5)
    assert torch.allclose(b1.grad, b2.grad, rtol=1e-5, atol=1e-4)
    # the GCN model: conv1's ReLU is fused; same numbers as ReLU applied outside
    torch.manual_seed(0)
    model = dnn.GCN(N, 7, 16).to(d)
    dcsr = dgsparse.SparseTensor(rowptr=A1.storage.rowptr(), col=A1.storage.col(), values=v1.detach(), has_value=True)
    out = model(dcsr, X1.detach())
    h = torch.relu(dgsparse.spmm_sum(dcsr, model.conv1.W(X1.detach()), 0))
    ref = dgsparse.spmm_sum(dcsr, model.conv2.W(h), 0)
    assert torch.equal(out, ref)
