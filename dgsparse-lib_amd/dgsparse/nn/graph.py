"""edge_index -> CSR helpers in plain torch (the reference leans on torch_sparse for these,
dgsparse/nn/gcnconv.py:36-49, ginconv.py:41-58; torch_sparse is not a dependency here)."""
from typing import Optional, Tuple

import torch


def csr_from_edge_index(edge_index: torch.Tensor, num_nodes: int, values: Optional[torch.Tensor] = None,
                        self_loops: Optional[float] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(rowptr int32, col int32, values float32) with entries sorted by (row, col) like ``torch_sparse``'s
    ``SparseTensor(row=edge_index[0], col=edge_index[1]).csr()``.  ``self_loops=w`` replaces the diagonal by w
    (``torch_sparse.fill_diag`` semantics).  Duplicate edges are kept as separate entries."""
    row, col = edge_index[0].long(), edge_index[1].long()
    if values is None:
        values = torch.ones(row.numel(), dtype=torch.float32, device=row.device)
    if self_loops is not None:
        keep = row != col
        diag = torch.arange(num_nodes, device=row.device)
        row = torch.cat([row[keep], diag])
        col = torch.cat([col[keep], diag])
        values = torch.cat([values[keep], torch.full((num_nodes,), float(self_loops), device=row.device)])
    order = torch.argsort(row * num_nodes + col, stable=True)
    row, col, values = row[order], col[order], values[order].to(torch.float32)
    rowptr = torch.zeros(num_nodes + 1, dtype=torch.int64, device=row.device)
    rowptr[1:] = torch.cumsum(torch.bincount(row, minlength=num_nodes), 0)
    return rowptr.to(torch.int32), col.to(torch.int32), values
