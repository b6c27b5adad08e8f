#!/usr/bin/env python3
"""End-to-end use of the operator surface: train a 2-layer GCN (dgsparse.nn.GCN) on a synthetic, dataset-shaped
graph and compare step time + loss trajectory with the same model written on torch.sparse.mm (hipSPARSE).
Every step runs SpMM forward, and in backward one SDDMM (grad of the edge weights) + one SpMM on the CSC arrays.
The counterpart of the reference's scratch scripts test/test_dgl.py / test/test_GIN.py, with assertions.

    python examples/train_gcn.py [--dataset pubmed] [--epochs 30] [--hidden 64]
"""
import argparse
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dgsparse-lib_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import dgsparse  # noqa: E402
from bench import graphgen  # noqa: E402
from dgsparse import nn as dnn  # noqa: E402

warnings.filterwarnings('ignore', message='Sparse CSR tensor support is in beta')


class TorchGCN(torch.nn.Module):
    def __init__(self, a, b, h):
        super().__init__()
        self.W1 = torch.nn.Linear(a, h, bias=False)
        self.W2 = torch.nn.Linear(h, b, bias=False)

    def forward(self, A, x):
        return torch.sparse.mm(A, self.W2(F.relu(torch.sparse.mm(A, self.W1(x)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dataset', default='pubmed')
    ap.add_argument('--epochs', type=int, default=30)
    ap.add_argument('--hidden', type=int, default=64)
    ap.add_argument('--feat', type=int, default=128)
    ap.add_argument('--classes', type=int, default=8)
    a = ap.parse_args()
    dev = 'cuda'
    rp, col, st = graphgen.dataset_shaped(a.dataset, seed=0, device=dev, as_torch=True)
    n = st['M']
    rows = torch.repeat_interleave(torch.arange(n, device=dev), (rp[1:] - rp[:-1]).long())
    ei = torch.stack([rows, col.long()])
    dcsr = dnn.get_gcn_dcsr_from_edge_index(ei, n)  # D^-1/2 (A+I) D^-1/2, values require grad
    s = dcsr.storage
    A_t = torch.sparse_csr_tensor(s.rowptr(), s.col(), s.values().detach(), size=(n, n))
    x = torch.rand(n, a.feat, device=dev)
    y = torch.randint(0, a.classes, (n,), device=dev)
    torch.manual_seed(0)
    m1 = dnn.GCN(a.feat, a.classes, a.hidden).to(dev)
    m2 = TorchGCN(a.feat, a.classes, a.hidden).to(dev)
    m2.W1.weight.data.copy_(m1.conv1.W.weight.data)
    m2.W2.weight.data.copy_(m1.conv2.W.weight.data)
    o1 = torch.optim.SGD(m1.parameters(), lr=0.5)
    o2 = torch.optim.SGD(m2.parameters(), lr=0.5)

    def run(model, opt, adj):
        losses = []
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(a.epochs):
            opt.zero_grad(set_to_none=True)
            loss = F.cross_entropy(model(adj, x), y)
            loss.backward()
            opt.step()
            losses.append(loss.detach())
        torch.cuda.synchronize()
        return (time.time() - t) / a.epochs * 1e3, torch.stack(losses).cpu()

    run(m1, o1, dcsr), run(m2, o2, A_t)  # warm-up (also moves both models identically)
    t1, l1 = run(m1, o1, dcsr)
    t2, l2 = run(m2, o2, A_t)
    print(f'{a.dataset}-shaped: {n} nodes, {st["nnz"]} edges (+self loops), feat {a.feat} -> {a.hidden} -> {a.classes}')
    print(f'dgsparse GCN   : {t1:8.3f} ms/epoch   loss {l1[0]:.5f} -> {l1[-1]:.5f}')
    print(f'torch.sparse.mm: {t2:8.3f} ms/epoch   loss {l2[0]:.5f} -> {l2[-1]:.5f}')
    assert torch.allclose(l1, l2, rtol=2e-3, atol=2e-4), 'loss trajectories diverged'
    assert dcsr.storage._values.grad is not None and torch.isfinite(dcsr.storage._values.grad).all()
    print('loss trajectories match; edge-weight gradients present')


if __name__ == '__main__':
    main()
