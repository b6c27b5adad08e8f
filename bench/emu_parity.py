#!/usr/bin/env python3
"""bench.py's `parity` block on the CPU EMULATION of the kernels (tests/emu) - for rounds without a GPU.

The headline graph of bench.py - since round 5 the SAME rowptr / col / values / features the GPU run times, bit for bit
(bench/graphgen.py: counter-based sampler) - through the emulated default sum, plan-free and planned, with the hub chains on
(the default, behind the device self-test) and off (round 3's schedule), every element against the reference's own sequential
host loop (oracle/_ref: spmm_reference_host; the C restatement without it) and a float64 sum.  Says what `parity.within_1e_5` of
the timed schedule will say on hardware as far as arithmetic ORDER decides it - the emulation executes the same source with the
same roundings; it says nothing about time.
    python bench/emu_parity.py [--rows-log2 20] [--seeds 0,1,2,3,4] [--feats 32,64,128] > profiles/r05_emu_parity.json
"""
import argparse
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
from bench import graphgen  # noqa: E402


def one(logm, seed, N, runs):
    M = 1 << logm
    rp, col, st = graphgen.powerlaw_csr(M, M * 16, alpha=2.1, dmax=1 << 16, cols='powerlaw', seed=seed, sampler='hash')
    val = graphgen.values_t(st['nnz'], seed).numpy()
    X = graphgen.features_t(st['K'], N, seed).numpy()
    lens = np.diff(rp)
    nthr = oracle.max_threads()
    if oracle.have_ref() and seed == 0 and N == 64:
        Cseq = np.asarray(oracle.ref_spmm_sum(rp, col, val, X)).reshape(M, N)  # the reference's own loop (single thread)
        refname = 'reference spmm_reference_host (oracle/_ref)'
    else:  # its restatement on all cores (pinned bit for bit to that loop by tests/test_oracle_pin.py)
        Cseq = oracle.spmm('sum', rp, col, val, X, fma=False, threads=nthr)[0]
        refname = 'oracle sequential fp32 (mul, add), OpenMP over rows'
    Cfma = oracle.spmm('sum', rp, col, val, X, fma=True, threads=nthr)[0]
    C64 = oracle.spmm_sum_f64(rp, col, val, X)
    e_seq = np.abs(Cseq - C64) / np.maximum(np.abs(C64), 1e-6)
    out = dict(graph=f'power-law CSR {M}x{st["K"]}, nnz {col.shape[0]}, longest row {int(lens.max())}, feat {N}, seed {seed}, '
                     'values / features U[0,1) (bench.py\'s tensors: graphgen sampler=hash)',
               reference=refname + ' - one sequential chain per (row, feature)',
               rows_gt_16384=int((lens > 16384).sum()), rows_gt_8192=int((lens > 8192).sum()),
               max_rel_err_vs_fp64_of_the_reference_itself=float(e_seq.max()), runs={})
    for name, env, planned in runs:
        E.set_env(DGS_HUB_CHAIN=None, DGS_NBU=None)
        E.set_env(**env)
        plan = E.spmm_plan(rp, col, st['K']) if planned else None
        t0 = time.time()
        C, _ = E.spmm(E.SUM, rp, col, val, X, plan=plan)
        dt = time.time() - t0
        rel = np.abs(C.astype(np.float64) - Cseq) / np.maximum(np.abs(Cseq), 1e-6)
        e_gpu = np.abs(C - C64) / np.maximum(np.abs(C64), 1e-6)
        th = 0 if 'DGS_HUB_CHAIN' in env else 16384
        hub = lens > th if th else np.zeros(M, bool)
        mid = (lens > 64) & (lens <= 16384)
        out['runs'][name] = dict(
            max_rel_err_vs_sequential=float(rel.max()), within_1e_5=bool(rel.max() <= 1e-5), elements_beyond_1e_5=int((rel > 1e-5).sum()),
            elements=int(rel.size), max_rel_err_rows_le_64nnz=float(rel[lens <= 64].max()),
            max_rel_err_rows_65_to_16384=float(rel[mid].max()) if mid.any() else None,
            max_rel_err_rows_gt_16384=float(rel[lens > 16384].max()) if (lens > 16384).any() else None,
            hub_rows_bit_exact_vs_fmaf_chain=bool(np.array_equal(C[hub].view(np.int32), Cfma[hub].view(np.int32))) if hub.any() else None,
            max_rel_err_vs_fp64=float(e_gpu.max()),
            n_hub_in_plan=(int(plan[1].n_hub) if plan is not None else None), emulation_seconds=round(dt, 1))
    E.set_env(DGS_HUB_CHAIN=None, DGS_NBU=None)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows-log2', type=int, default=20)
    ap.add_argument('--seeds', default='0')
    ap.add_argument('--feats', default='64')
    ap.add_argument('--hub-off', action='store_true', help='also run the schedule without hub chains (round 3\'s)')
    a = ap.parse_args()
    runs = [('hub chains on (default), plan-free', {}, False), ('hub chains on (default), planned', {}, True)]
    if a.hub_off:
        runs.append(('hub chains off (round-3 schedule), planned', dict(DGS_HUB_CHAIN=0), True))
    res = dict(emulation='tests/emu (wave64 emulation of dgsparse-lib_amd/csrc, same C ABI, 256 CUs)',
               hub_self_test_on_the_emulation=int(E.lib().dgs_spmm_hub_gate()), cells={})
    for seed in (int(x) for x in a.seeds.split(',')):
        for N in (int(x) for x in a.feats.split(',')):
            cell = one(a.rows_log2, seed, N, runs)
            res['cells'][f'seed{seed}_feat{N}'] = cell
            print(f'seed {seed} feat {N}: ' + '; '.join(f'{k}: max {v["max_rel_err_vs_sequential"]:.3e} beyond {v["elements_beyond_1e_5"]}'
                                                        for k, v in cell['runs'].items()), file=sys.stderr, flush=True)
    res['all_cells_within_1e_5'] = all(r['within_1e_5'] for c in res['cells'].values() for k, r in c['runs'].items() if 'off' not in k)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
