"""What the hub chains of round 4 cost in L2 misses, by the LRU model of lru_streams.py (a fully associative 16 k-row LRU per
XCD fed with the reference stream of the fused launch; it reproduced the measured hit rates of rounds 2 / 3 to a point).
XCD 0's stream = its eighth of the short rows + the units of column slice 0 (rows > 256 nnz, cut on the grid, in column order)
+ its share of the uncut 65 .. 256-nnz rows, interleaved in proportion.  With hub chains, rows longer than the threshold leave
the slice stream (on EVERY XCD) and the hub rows dealt to XCD 0 (every eighth, longest first) walk ALL their columns there, in
column order, at a pace that ends with the launch.     python experiments/l2_model/hub_chain_cost.py   (CPU, ~2 min)"""
import os
import sys
from collections import OrderedDict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import graphgen  # noqa: E402

rp, col, st = graphgen.powerlaw_csr(1 << 20, 1 << 24, seed=0)
M, K, nnz = st['M'], st['K'], st['nnz']
lens = np.diff(rp)
cnt = np.bincount(col, minlength=K)
cum = np.cumsum(cnt)
bounds = [0] + [int(np.searchsorted(cum, nnz * x / 8)) for x in range(1, 8)] + [K]
row_of = np.repeat(np.arange(M), lens)
CAP = 16384


def lru_misses(stream):
    od, miss = OrderedDict(), 0
    for c in stream.tolist():
        if c in od:
            od.move_to_end(c)
        else:
            miss += 1
            od[c] = 1
            if len(od) > CAP:
                od.popitem(last=False)
    return miss


def interleave(arrs):
    arrs = [a for a in arrs if len(a)]
    keys = np.concatenate([np.arange(len(a)) / len(a) for a in arrs])
    return np.concatenate(arrs)[np.argsort(keys, kind='stable')]


def unit_stream(sel):
    """nnz selected by `sel` (inside slice 0), as units of <= 256 nnz per row, units in column order, 512 in flight."""
    idx = np.nonzero(sel)[0]
    r = row_of[idx]
    first = np.r_[True, r[1:] != r[:-1]]
    segstart = np.maximum.accumulate(np.where(first, np.arange(len(idx)), 0))
    k = (np.arange(len(idx)) - segstart) // 256
    ukey = r * 1000 + k
    ufirst = np.r_[True, ukey[1:] != ukey[:-1]]
    uid = np.cumsum(ufirst) - 1
    order_units = np.argsort(col[idx][ufirst], kind='stable')
    pos = np.empty(len(order_units), np.int64)
    pos[order_units] = np.arange(len(order_units))
    upos = pos[uid]
    within = np.arange(len(idx)) - np.maximum.accumulate(np.where(ufirst, np.arange(len(idx)), 0))
    key = (upos // 512) * 100000 + (within // 4) * 600 + (upos % 512)
    return col[idx][np.argsort(key, kind='stable')]


short = lens[row_of] <= 64
s_rows = col[(row_of < M // 8) & short]
mid = (lens[row_of] > 64) & (lens[row_of] <= 256) & ((row_of * 2654435761 % 8) == 0)
s_mid = col[mid]
in0 = col < bounds[1]
base = None
for th in (0, 32768, 16384, 8192, 4096):
    sliced = (lens[row_of] > 256) & in0 & ((lens[row_of] <= th) if th else True)
    s_units = unit_stream(sliced)
    streams = [s_rows, s_units, s_mid]
    hub_nnz = 0
    if th:
        hub_rows = np.nonzero(lens > th)[0]
        hub_rows = hub_rows[np.argsort(-lens[hub_rows], kind='stable')][0::8]  # XCD 0's deal: every eighth, longest first
        # the hub rows of an XCD run concurrently, each at a pace proportional to its length
        hs = [col[rp[r]:rp[r + 1]] for r in hub_rows]
        if hs:
            streams.append(interleave(hs))
            hub_nnz = sum(len(h) for h in hs)
    s_all = interleave(streams)
    miss = lru_misses(s_all)
    if base is None:
        base = miss
    print(f'hub threshold {th or "off":>6}: XCD 0 references {len(s_all)} (hub rows here: {hub_nnz} nnz), misses {miss} '
          f'({miss / len(s_all):.4f} of the references), {miss / base:.4f} x the misses without hub chains', flush=True)
