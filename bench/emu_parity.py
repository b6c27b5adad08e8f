#!/usr/bin/env python3
"""bench.py's `parity` block on the CPU EMULATION of the kernels (tests/emu) - for rounds without a GPU.

A scaled-down headline graph (2^17 rows, ~16 nnz/row, the same degree / column laws and value distributions as bench.py, rows up
to ~5 10^4 nnz) through the emulated default sum with the hub chains on (plan-free and planned) and off, every element against
the reference's own sequential host loop (oracle/_ref: spmm_reference_host; the C restatement without it) and a float64 sum.
Says what the timed schedule's `parity.within_1e_5` will say on hardware as far as arithmetic ORDER decides it - the emulation
executes the same source with the same roundings; it says nothing about time.    python bench/emu_parity.py > profiles/r04_emu_parity.json
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
from bench import graphgen  # noqa: E402


def main():
    logm = int(sys.argv[1]) if len(sys.argv) > 1 else 17
    M, N = 1 << logm, 64
    rp, col, st = graphgen.powerlaw_csr(M, M * 16, alpha=2.1, dmax=1 << 16, cols='powerlaw', seed=0)
    rng = np.random.default_rng(1)
    val = rng.random(col.shape[0], dtype=np.float32)
    X = rng.random((st['K'], N), dtype=np.float32)
    lens = np.diff(rp)
    Cseq = oracle.ref_spmm_sum(rp, col, val, X) if oracle.have_ref() else oracle.spmm('sum', rp, col, val, X, threads=oracle.max_threads())[0]
    Cseq = np.asarray(Cseq).reshape(M, N)
    Cfma = oracle.spmm('sum', rp, col, val, X, fma=True, threads=oracle.max_threads())[0]
    C64 = oracle.spmm_sum_f64(rp, col, val, X)
    out = dict(graph=f'power-law CSR {M}x{st["K"]}, nnz {col.shape[0]}, longest row {int(lens.max())}, feat {N}, values / features U[0,1)',
               reference=('reference spmm_reference_host (oracle/_ref)' if oracle.have_ref() else 'oracle sequential fp32') +
               ' - one sequential chain per (row, feature)', emulation='tests/emu (wave64 emulation of dgsparse-lib_amd/csrc, same C ABI)',
               rows_gt_16384=int((lens > 16384).sum()), rows_gt_8192=int((lens > 8192).sum()), runs={})
    for name, env, planned in (('hub chains on (default), plan-free', {}, False), ('hub chains on (default), planned', {}, True),
                               ('hub chains off (round-3 schedule), planned', dict(DGS_HUB_CHAIN=0), True)):
        E.set_env(DGS_HUB_CHAIN=None, DGS_NBU=64)
        E.set_env(**env)
        plan = E.spmm_plan(rp, col, st['K']) if planned else None
        t0 = time.time()
        C, _ = E.spmm(E.SUM, rp, col, val, X, plan=plan)
        dt = time.time() - t0
        rel = np.abs(C.astype(np.float64) - Cseq) / np.maximum(np.abs(Cseq), 1e-6)
        e_gpu = np.abs(C - C64) / np.maximum(np.abs(C64), 1e-6)
        e_seq = np.abs(Cseq - C64) / np.maximum(np.abs(C64), 1e-6)
        th = 0 if 'DGS_HUB_CHAIN' in env else 16384
        hub = lens > th if th else np.zeros(M, bool)
        out['runs'][name] = dict(
            max_rel_err_vs_sequential=float(rel.max()), within_1e_5=bool(rel.max() <= 1e-5), elements_beyond_1e_5=int((rel > 1e-5).sum()),
            elements=int(rel.size), max_rel_err_rows_le_64nnz=float(rel[lens <= 64].max()),
            max_rel_err_rows_65_to_16384=float(rel[(lens > 64) & (lens <= 16384)].max()),
            max_rel_err_rows_gt_16384=float(rel[lens > 16384].max()) if (lens > 16384).any() else None,
            hub_rows_bit_exact_vs_fmaf_chain=bool(np.array_equal(C[hub].view(np.int32), Cfma[hub].view(np.int32))) if hub.any() else None,
            max_rel_err_vs_fp64=float(e_gpu.max()), max_rel_err_vs_fp64_of_the_reference_itself=float(e_seq.max()),
            n_hub_in_plan=(int(plan[1].n_hub) if plan is not None else None), emulation_seconds=round(dt, 1))
    E.set_env(DGS_HUB_CHAIN=None, DGS_NBU=None)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
