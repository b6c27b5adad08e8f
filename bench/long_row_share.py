#!/usr/bin/env python3
"""Share of rows and of nnz above a hub threshold, for every BASELINE.json config (VERDICT r5 #4b) - CPU only, from the generators'
DEGREE law (bench/graphgen.powerlaw_degrees with the (M, nnz, dmax, alpha) of graphgen.SHAPES, seeds 0 .. 4; the column draw and
its de-duplication change a row's length by a fraction of a percent and are not run here: Reddit-shaped has 114 M nnz).
What a lower DGS_HUB_CHAIN would chain: every row above it becomes ONE sequential chain per feature (a critical path of ~4 ns per
nnz on its own workgroups) and leaves the plan's column-slice order.
    python bench/long_row_share.py > profiles/r06_long_row_share.json        (~1 min)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import graphgen  # noqa: E402

THRESHOLDS = (1024, 2048, 4096, 16384)
CONFIGS = [('C1 cora-shaped', 'cora', 1.0), ('C2 arxiv-shaped', 'arxiv', 1.0), ('C3 reddit-shaped', 'reddit', 1.0),
           ('C4 products-shaped (SDDMM: no chains; its SpMM twin)', 'products', 1.0), ('headline synth 1M x 1M', 'synth1m', 1.0),
           ('C5 shard: 2^21 rows of synth 16M (one GPU of 8)', 'synth16m', 0.125)]


def main():
    out = {}
    for label, name, frac in CONFIGS:
        s = graphgen.SHAPES[name]
        M, nnz = int(s['M'] * frac), int(s['nnz'] * frac)
        per = {str(t): dict(rows=[], nnz_share=[]) for t in THRESHOLDS}
        longest = []
        for seed in range(5):
            rng = np.random.Generator(np.random.PCG64(seed))
            deg = graphgen.powerlaw_degrees(M, nnz, s['alpha'], min(M, s['dmax']), rng)
            longest.append(int(deg.max()))
            for t in THRESHOLDS:
                m = deg > t
                per[str(t)]['rows'].append(int(m.sum()))
                per[str(t)]['nnz_share'].append(float(deg[m].sum() / deg.sum()))
        out[label] = dict(M=M, nnz=nnz, alpha=s['alpha'], dmax=s['dmax'], longest_row=dict(min=min(longest), max=max(longest)),
                          above={t: dict(rows_min=min(v['rows']), rows_max=max(v['rows']),
                                         nnz_share_mean=round(float(np.mean(v['nnz_share'])), 5),
                                         longest_chain_us_at_4ns_per_nnz=round(max(longest) * 4e-3, 1) if max(longest) > int(t) else 0.0)
                                 for t, v in per.items()})
    json.dump(out, sys.stdout, indent=1)
    print()
    print('\n| config | rows / nnz | longest row | > 1 024 | > 2 048 | > 4 096 | > 16 384 |', file=sys.stderr)
    print('|---|---|---|---|---|---|---|', file=sys.stderr)
    for label, v in out.items():
        cells = ' | '.join(f"{a['rows_min']}-{a['rows_max']} rows, {100 * a['nnz_share_mean']:.1f} % nnz" for a in v['above'].values())
        print(f"| {label} | {v['M']} / {v['nnz']} | {v['longest_row']['min']}-{v['longest_row']['max']} | {cells} |", file=sys.stderr)


if __name__ == '__main__':
    main()
