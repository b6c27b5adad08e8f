"""Synthetic CSR generators (pure numpy) shared by tests/ and bench.py.

No dataset can be downloaded here or on the GPU box, so every BASELINE.json config is
realised as a seeded synthetic CSR with the dataset's shape: rows, nnz, degree law.  Column ids are
sorted inside each row (the scipy ``tocsr`` convention the reference tests rely on,
/root/reference/test/utils.py:47-49); duplicates are removed unless ``dedup=False``.
"""
from __future__ import annotations

import numpy as np

# (M, nnz, max_degree, alpha) of the graphs BASELINE.json names; shapes from SURVEY.md 8(a)/(d).
SHAPES = {
    'cora': dict(M=2708, nnz=10556, dmax=168, alpha=2.6),
    'citeseer': dict(M=3327, nnz=9104, dmax=99, alpha=2.8),
    'pubmed': dict(M=19717, nnz=88648, dmax=171, alpha=2.6),
    'ppi': dict(M=56944, nnz=1612348, dmax=721, alpha=2.9),
    'ppi0': dict(M=1767, nnz=32318, dmax=200, alpha=2.9),  # graph 0 of PPI = what test/utils.py:36-38 loads
    'arxiv': dict(M=169343, nnz=1166243, dmax=13161, alpha=2.1),
    'reddit': dict(M=232965, nnz=114615892, dmax=21657, alpha=3.5),
    'products': dict(M=2449029, nnz=61859140, dmax=17481, alpha=2.4),
    'synth1m': dict(M=1 << 20, nnz=1 << 24, dmax=1 << 16, alpha=2.1),
    'synth16m': dict(M=1 << 24, nnz=1 << 29, dmax=1 << 17, alpha=2.1),
}


def powerlaw_degrees(M: int, nnz: int, alpha: float, dmax: int, rng) -> np.ndarray:
    """Row degrees ~ truncated Pareto (tail exponent alpha-1), rescaled so that sum == nnz.

    Rows with degree 0 appear naturally (they exercise the empty-row rule, spmm_cuda.cuh:49-51).
    """
    u = rng.random(M)
    raw = (1.0 - u) ** (-1.0 / (alpha - 1.0))  # Pareto, xmin = 1
    raw = np.minimum(raw, float(dmax))
    for _ in range(8):  # fixed-point rescale under the dmax clip
        scale = nnz / raw.sum()
        raw = np.minimum(raw * scale, float(dmax))
    deg = np.floor(raw + rng.random(M)).astype(np.int64)  # stochastic rounding keeps the mean
    deg = np.minimum(deg, dmax)
    diff = int(nnz - deg.sum())
    if diff != 0:  # spread the remainder over random rows
        idx = rng.integers(0, M, size=abs(diff))
        np.add.at(deg, idx, 1 if diff > 0 else -1)
        deg = np.clip(deg, 0, None)
    return deg


def hash_bits(n: int, seed: int, stream: int, device='cpu', start: int = 0):
    """n pseudo-random int64 words, word i = splitmix64(seed, stream, start + i): a COUNTER-based generator written with integer
    tensor ops only, so the CPU and the GPU produce the same words (torch's own generators differ per device type: the round-4
    emulation could only reproduce a sibling of the graph bench.py timed on the GPU, VERDICT r4 #2b)."""
    import torch

    def lsr(x, k):  # logical shift right of an int64 tensor
        return (x >> k) & ((1 << (64 - k)) - 1)

    def c(v):  # 64-bit constant as a signed int64 (multiplication wraps)
        return v - (1 << 64) if v >= (1 << 63) else v

    z = torch.arange(start, start + n, dtype=torch.int64, device=device)
    z = (z + c((seed * 0xD1B54A32D192ED03 + stream * 0x8CB92BA72F3D8DD7 + 0x9E3779B97F4A7C15) & ((1 << 64) - 1))) * c(0x9E3779B97F4A7C15)
    z = (z ^ lsr(z, 30)) * c(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * c(0x94D049BB133111EB)
    return z ^ lsr(z, 31)


def hash_unit(n: int, seed: int, stream: int, device='cpu', dtype=None, start: int = 0):
    """n numbers in [0, 1) from hash_bits: float64 with 53 random bits, or float32 with 24 (exact conversions: the same values
    on every device)."""
    import torch
    b = hash_bits(n, seed, stream, device, start)
    if dtype in (None, torch.float64):
        return ((b >> 11) & ((1 << 53) - 1)).to(torch.float64) * (1.0 / (1 << 53))
    assert dtype == torch.float32
    return ((b >> 40) & ((1 << 24) - 1)).to(torch.float32) * (1.0 / (1 << 24))


def values_t(nnz: int, seed: int, device='cpu'):
    """Edge values U[0,1) float32 as a torch tensor on `device`, the same bits on CPU and GPU (what bench.py times)."""
    return hash_unit(nnz, seed, 101, device, __import__('torch').float32)


def features_t(K: int, N: int, seed: int, device='cpu'):
    """Dense operand U[0,1) float32 [K, N] as a torch tensor on `device`, the same bits on CPU and GPU."""
    return hash_unit(K * N, seed, 102, device, __import__('torch').float32).view(K, N)


def _sample_cols(deg_t, K, cols, cdf, gen, device, stream=1):
    """`gen`: a torch.Generator (the historical streams, different on CPU and GPU) or an int seed (hash_bits: device-independent)."""
    import torch
    total = int(deg_t.sum())
    row = torch.repeat_interleave(torch.arange(deg_t.shape[0], device=device), deg_t)
    hashed = isinstance(gen, int)
    if cols == 'uniform':
        col = (hash_bits(total, gen, stream, device) >> 1 & ((1 << 62) - 1)) % K if hashed else \
            torch.randint(0, K, (total,), generator=gen, device=device)
    elif cols == 'local':  # community-like: columns within a window around the row id (graphs in a locality-
        w = max(64, K // 256)  # preserving order, e.g. after METIS/RCM reordering)
        off = ((hash_bits(total, gen, stream, device) >> 1 & ((1 << 62) - 1)) % (2 * w + 1) - w) if hashed else \
            torch.randint(-w, w + 1, (total,), generator=gen, device=device)
        col = (row * K // deg_t.shape[0] + off).clamp_(0, K - 1)
    else:
        u = hash_unit(total, gen, stream, device) if hashed else torch.rand(total, generator=gen, device=device, dtype=torch.float64)
        col = torch.searchsorted(cdf, u, right=True).clamp_(max=K - 1)
    return row * K + col


def powerlaw_csr(M: int, nnz: int, K: int | None = None, alpha: float = 2.1, dmax: int | None = None,
                 cols: str = 'powerlaw', seed: int = 0, dedup: bool = True, device: str = 'cpu',
                 as_torch: bool = False, sampler: str = 'torch'):
    """Power-law CSR.  ``cols``: 'powerlaw' = column popularity follows the same degree law
    (Chung-Lu style: hubs are both prolific and popular, as in citation/social graphs);
    'uniform' = columns uniform in [0,K) (worst case for cache reuse of the dense operand).
    The big arrays are built with torch on ``device`` (a GPU when bench.py has one; streams differ
    per device type but are seeded; ``sampler='hash'``: the counter-based generator above instead - the SAME graph whatever the
    device, which is what bench.py times since round 5).  Duplicates are removed and topped up once, so nnz' ~= nnz.
    Returns (rowptr int32 [M+1], col int32 [nnz'], stats dict)."""
    import torch
    rng = np.random.Generator(np.random.PCG64(seed))
    K = M if K is None else K
    dmax = min(K, dmax if dmax is not None else max(1, M // 16))
    deg = powerlaw_degrees(M, nnz, alpha, dmax, rng)
    if sampler == 'hash':
        gen = int(seed)
    else:
        gen = torch.Generator(device=device)
        gen.manual_seed(seed)
    cdf = None
    if cols == 'powerlaw':
        w = powerlaw_degrees(K, nnz, alpha, dmax, rng).astype(np.float64) + 0.05
        w = w[rng.permutation(K)]  # popularity uncorrelated with the row id
        c = np.cumsum(w)
        cdf = torch.from_numpy(c / c[-1]).to(device)
    elif cols not in ('uniform', 'local'):
        raise ValueError(cols)
    deg_t = torch.from_numpy(deg).to(device)
    key = _sample_cols(deg_t, K, cols, cdf, gen, device)
    key = torch.sort(key).values
    if dedup:
        key = torch.unique_consecutive(key)
        have = torch.bincount(key // K, minlength=M)
        lack = (deg_t - have).clamp_(min=0)
        lack = torch.minimum(lack, K - have)
        if int(lack.sum()) > 0:  # one top-up round for rows that lost duplicates
            extra = _sample_cols(lack, K, cols, cdf, gen, device, stream=2)
            key = torch.unique_consecutive(torch.sort(torch.cat([key, extra])).values)
    row = key // K
    col = (key - row * K).to(torch.int32)
    counts = torch.bincount(row, minlength=M)
    rowptr = torch.zeros(M + 1, dtype=torch.int64, device=device)
    torch.cumsum(counts, 0, out=rowptr[1:])
    assert int(rowptr[-1]) < 2**31
    stats = dict(M=M, K=K, nnz=int(rowptr[-1]), max_deg=int(counts.max()),
                 empty_rows=int((counts == 0).sum()), cols=cols, alpha=alpha, seed=seed, gen_device=device, sampler=sampler)
    rowptr = rowptr.to(torch.int32)
    if as_torch:
        return rowptr, col, stats
    return rowptr.cpu().numpy(), col.cpu().numpy(), stats


def dataset_shaped(name: str, seed: int = 0, cols: str = 'powerlaw', scale: float = 1.0, **kw):
    """CSR with the (M, nnz, max degree) of a named dataset.  ``scale`` < 1 shrinks M and nnz."""
    s = SHAPES[name]
    M = max(8, int(s['M'] * scale))
    nnz = max(8, int(s['nnz'] * scale))
    return powerlaw_csr(M, nnz, alpha=s['alpha'], dmax=min(M, s['dmax']), cols=cols, seed=seed, **kw)


def features(K: int, N: int, seed: int = 0) -> np.ndarray:
    """Dense operand U[0,1) float32, as /root/reference/test/test_spmm.py:20."""
    rng = np.random.Generator(np.random.PCG64(seed + 1000))
    return rng.random((K, N), dtype=np.float32)


def weights(nnz: int, kind: str = 'ones', seed: int = 0) -> np.ndarray:
    """'ones' (test/utils.py:52), 'uniform' U[0,1), 'tied' {0,.1,.2} (sp_util.hpp:44-48), 'signed'."""
    rng = np.random.Generator(np.random.PCG64(seed + 2000))
    if kind == 'ones':
        return np.ones(nnz, np.float32)
    if kind == 'uniform':
        return rng.random(nnz, dtype=np.float32)
    if kind == 'tied':
        return (rng.integers(0, 3, nnz) / 10).astype(np.float32)
    if kind == 'signed':
        return (rng.random(nnz, dtype=np.float32) - 0.5).astype(np.float32)
    raise ValueError(kind)
