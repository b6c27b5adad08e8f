"""GCN / GIN layers: forward + backward through SpMM + SDDMM + csr2csc against a dense torch
computation of the same model on a small graph."""
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
