#!/usr/bin/env python3
"""Discrete-event estimate (CPU, no GPU) of WHEN the hub rows of the headline graph are finished under the two hub modes - not a
measurement, a plausibility check of the pipeline the slice-by-slice mode (DGS_HUB_XCD=1) sets up.

Inputs that ARE measured (profiles/r04_lds_dma_gather.txt part 2, fabric saturated): a hub workgroup with two gather sets in flight
chains a link in ~4 ns when its gathers miss; ASSUMED: 3 ns when two thirds of them hit the slice's L2, 3 us per hand-over
(device-scope store -> poll on another XCD), 4 feature slices per row, the launch's hub grid (one workgroup per task up to
4 per CU, a multiple of 8).  Output: finish time of the last hub task and the idle share of the hub workgroups, both modes."""
import heapq
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import graphgen  # noqa: E402

NS_MISS, NS_HIT, HANDOVER_US, SH, TH = 4.0, 3.0, 3.0, 4, 16384
M = 1 << 20
rp, col, st = graphgen.powerlaw_csr(M, M * 16, alpha=2.1, dmax=1 << 16, cols='powerlaw', seed=0)
lens = np.diff(rp)
rows = np.nonzero(lens > TH)[0]
rows = rows[np.argsort(-lens[rows], kind='stable')]
# the plan's 8 column slices: equal reference counts
cnt = np.bincount(col, minlength=st['K'])
cum = np.cumsum(cnt)
bounds = [0] + [int(np.searchsorted(cum, cum[-1] * x / 8)) for x in range(1, 8)] + [st['K']]
seg = np.array([[np.count_nonzero((col[rp[r]:rp[r + 1]] >= bounds[s]) & (col[rp[r]:rp[r + 1]] < bounds[s + 1])) for s in range(8)]
                for r in rows])
ntask = len(rows) * SH
nbh = min((ntask + 7) // 8 * 8, 1024)
print(f'{len(rows)} hub rows, {int(lens[rows].sum())} nnz, longest {int(lens[rows[0]])}; {ntask} tasks, {nbh} hub workgroups')
print('nnz per column slice of the longest row:', seg[0].tolist())

# one workgroup per row slice (whole row on one XCD; boustrophedon deal ignored: one task per workgroup when nbh >= ntask)
per = nbh // 8
t_whole = np.zeros(nbh)
for k in range(ntask):
    g = k // SH
    t_whole[k % nbh] += lens[rows[g]] * NS_MISS * 1e-3
print(f'one workgroup per (row, feature slice): last hub task done at {t_whole.max():.0f} us; busy share of the hub workgroups until then '
      f'{t_whole.sum() / (t_whole.max() * nbh):.2f}')

# slice by slice: slot p of XCD s works segment s of tasks p, p + per, ...; waits for XCD s - 1's same task
for nb in (nbh, 104, 64, 32):
    per = nb // 8
    done = np.zeros((8, ntask))
    free = np.zeros((8, per))
    busy = 0.0
    for k in range(ntask):  # (tasks in increasing order: a slot's i-th task only depends on lower k and lower s)
        g, p = k // SH, k % per
        for s in range(8):
            ready = done[s - 1, k] + HANDOVER_US if s else 0.0
            start = max(free[s, p], ready)
            dur = seg[g, s] * NS_HIT * 1e-3
            done[s, k] = start + dur
            free[s, p] = done[s, k]
            busy += dur
    last = done[7].max()
    print(f'slice by slice across the XCDs, {nb:4d} hub workgroups: last hub task done at {last:.0f} us; busy share until then '
          f'{busy / (last * nb):.2f}; workgroup-microseconds held {last * nb / 1000:.0f} k; the longest row alone: {done[7, :SH].max():.0f} us')
print(f'(one workgroup per task holds {t_whole.sum() / 1000:.0f} k workgroup-microseconds, all of them busy)')

# work-conserving: a workgroup of XCD s claims any unclaimed task whose segment s-1 is done; even slots scan longest-first, odd slots shortest-first
def sim_claim(nb, mixed=True):
    per = nb // 8
    done = np.full((8, ntask), np.inf); claimed = np.zeros((8, ntask), bool)
    ev = [(0.0, s, p) for s in range(8) for p in range(per)]  # (time a workgroup becomes free, xcd, slot)
    heapq.heapify(ev); busy = 0.0; held_until = np.zeros((8, per)); last = 0.0
    pending = 8 * ntask
    while ev and pending:
        t, s, p = heapq.heappop(ev)
        order = range(ntask) if (not mixed or p % 2 == 0) else range(ntask - 1, -1, -1)
        pick = None
        for k in order:
            if not claimed[s, k] and (s == 0 or done[s - 1, k] + HANDOVER_US <= t):
                pick = k; break
        if pick is None:
            if claimed[s].all():
                held_until[s, p] = t; continue
            # sleep until the next producer finishes
            nxt = min((done[s - 1, k] + HANDOVER_US for k in range(ntask) if not claimed[s, k] and np.isfinite(done[s - 1, k])), default=t + 1.0)
            heapq.heappush(ev, (max(nxt, t + 0.5), s, p)); continue
        claimed[s, pick] = True; pending -= 1
        dur = seg[pick // SH, s] * NS_HIT * 1e-3
        done[s, pick] = t + dur; busy += dur; last = max(last, t + dur); held_until[s, p] = t + dur
        heapq.heappush(ev, (t + dur, s, p))
    return last, busy / held_until.sum(), held_until.sum() / 1000
for nb in (nbh, 104, 64):
    for mixed in (False, True):
        print('claim', 'front/back' if mixed else 'longest first', nb, 'last %.0f us, busy %.2f, held %.0f k' % sim_claim(nb, mixed))
