#!/usr/bin/env python3
"""Worst-case probe of the 8 k .. 16 k-nnz band (VERDICT r4 #2c) on the CPU emulation of the kernels.

Rows above DGS_HUB_CHAIN = 16384 nnz are chained (bit-exact); rows up to 64 nnz are chains anyway; in between the default sum
is a fixed tree whose distance from the reference's sequential chain is the chain's OWN rounding drift (~sqrt(len) ulp).  The
headline graph has ~60 rows in the top half of that band; this matrix has THOUSANDS of them - 700 rows each of exactly 16384,
12288 and 8192 nnz (uniform random columns, sorted) plus short filler rows - with the two value laws the reference uses:
U[0,1) (test/test_spmm.py:20) and {0, .1, .2} (example/util/sp_util.hpp:44-48).  Every element of the emulated default sum,
plan-free and planned, against the sequential fp32 chain (mul, add: the reference's host loop; and fmaf: its kernel).
    python bench/band_probe.py [rows_per_class] > profiles/r05_band_probe.json
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'emu')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

import emu_lib as E  # noqa: E402
import oracle  # noqa: E402


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 700
    N, K = 64, 1 << 20
    rng = np.random.default_rng(2026)
    classes = (16384, 12288, 8192)
    M = 70000
    deg = rng.integers(0, 4, M)
    for c, L in enumerate(classes):
        deg[100 + c * per:100 + (c + 1) * per] = L
    rp = np.zeros(M + 1, np.int64)
    rp[1:] = np.cumsum(deg)
    assert rp[-1] < 2**31
    rp = rp.astype(np.int32)
    nnz = int(rp[-1])
    col = rng.integers(0, K, nnz, dtype=np.int64)
    key = np.repeat(np.arange(M, dtype=np.int64), deg) * K + col
    key.sort()
    col = (key % K).astype(np.int32)
    del key
    X = rng.random((K, N), dtype=np.float32)
    nthr = oracle.max_threads()
    E.set_env(DGS_PANEL=0)  # (368 nnz per row would take the column-panel sweep; this probe is about the row-stream schedule's tree)
    assert E.schedule(E.SUM, M, K, N, nnz) == 'rows'
    out = dict(matrix=f'{M} rows, {nnz} nnz: {per} rows each of exactly {classes} nnz, the rest 0 .. 3; {K} columns (uniform, sorted), feat {N}',
               hub_threshold=int(E.lib().dgs_spmm_hub_threshold()), emulation='tests/emu, 256 CUs', cases={})
    for vname in ('uniform U[0,1)', 'tied {0,.1,.2}'):
        if vname.startswith('uniform'):
            val = rng.random(nnz, dtype=np.float32)
            Xv = X
        else:
            val = (rng.integers(0, 3, nnz) / 10).astype(np.float32)
            Xv = (rng.integers(0, 3, (K, N)) / 10).astype(np.float32)
        seq = oracle.spmm('sum', rp, col, val, Xv, fma=False, threads=nthr)[0]
        fma = oracle.spmm('sum', rp, col, val, Xv, fma=True, threads=nthr)[0]
        plan = E.spmm_plan(rp, col, K)
        for sname, kw in (('plan-free', {}), ('planned', dict(plan=plan))):
            t0 = time.time()
            C, _ = E.spmm(E.SUM, rp, col, val, Xv, **kw)
            dt = time.time() - t0
            cell = dict(emulation_seconds=round(dt, 1))
            for rname, ref in (('vs sequential (mul, add) = reference host loop', seq), ('vs sequential fmaf = reference kernel', fma)):
                rel = np.abs(C.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-6)
                per_class = {}
                for L in classes:
                    m = deg == L
                    per_class[str(L)] = dict(max_rel_err=float(rel[m].max()), elements=int(rel[m].size),
                                             beyond_1e_5=int((rel[m] > 1e-5).sum()), p999=float(np.quantile(rel[m], 0.999)))
                cell[rname] = dict(max_rel_err=float(rel.max()), beyond_1e_5=int((rel > 1e-5).sum()), by_row_length=per_class)
            out['cases'][f'{vname}, {sname}'] = cell
            print(vname, sname, {k: (v['max_rel_err'], v['beyond_1e_5']) for k, v in cell.items() if isinstance(v, dict)}, file=sys.stderr, flush=True)
    out['zero_elements_beyond_1e_5'] = all(v[k]['beyond_1e_5'] == 0 for v in out['cases'].values() for k in v if isinstance(v[k], dict))
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
