#!/usr/bin/env python3
"""The reference's own published benchmark on its own matrices.

/root/reference/example/README.md:47-60 publishes GE-SpMM on example/data/p2p-Gnutella31.mtx (62 586^2, nnz 147 892, N = 32) on a
V100: 0.0450 - 0.857 ms over its algorithms, best 210.6 GFLOP/s, cuSPARSE 124.5 GFLOP/s; the driver that printed it is
example/ge-spmm/spmm.cu:145-215 (values and dense operand in {0, .1, .2}, 10 warm-up + 100 timed launches between events,
GFLOP/s = 2 nnz N / t).  This script runs that protocol on MI355X:

  * examples/spmm_mtx (the C-ABI twin of spmm.cu / sddmm.cu: no Python in the timed loop) on a MatrixMarket file written from
    the committed CSR fixtures (tests/golden/p2p_gnutella31_csr2csc.npz, ca_condmat_csr.npz: the reference's data files are
    not on the GPU box) for N = 32 / 64 / 128: sum / max / min / mean, plan-free and planned where a plan applies, strict
    order, SDDMM - each checked against a host loop by the driver itself;
  * torch.sparse.mm on the same CSR (hipSPARSE: the cuSPARSE row of the reference's table) with the same 10 + 100 protocol.

    python bench/mtx_bench.py [--out profiles/r04_mtx] -> one JSON per matrix
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dgsparse-lib_amd')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

MATRICES = {
    'p2p-Gnutella31': os.path.join(ROOT, 'tests', 'golden', 'p2p_gnutella31_csr2csc.npz'),
    'ca-CondMat': os.path.join(ROOT, 'tests', 'golden', 'ca_condmat_csr.npz'),
}
PUBLISHED_V100 = {  # example/README.md:47-60, N = 32, p2p-Gnutella31
    'cusparse_ms': 0.076032, 'cusparse_gflops': 124.49, 'gespmm_best_ms': 0.044950, 'gespmm_best_gflops': 210.57,
    'gespmm_alg0_ms': 0.045675, 'gespmm_alg0_gflops': 207.23, 'gespmm_worst_ms': 0.857259,
}


def write_mtx(path, rowptr, col, shape):
    rows = np.repeat(np.arange(shape[0], dtype=np.int64), np.diff(rowptr))
    with open(path, 'w') as f:
        f.write('%%MatrixMarket matrix coordinate pattern general\n')
        f.write(f'{shape[0]} {shape[1]} {col.shape[0]}\n')
        np.savetxt(f, np.stack([rows + 1, col.astype(np.int64) + 1], 1), fmt='%d %d')


def run_driver(mtx, N):
    exe = os.path.join(ROOT, 'examples', 'spmm_mtx')
    out = subprocess.run([exe, mtx, str(N)], capture_output=True, text=True, timeout=600)
    res = {'returncode': out.returncode}
    for line in out.stdout.splitlines():
        m = re.match(r'\[(.+?)\] (?:check|verification|bit-exact vs the sequential host loop:) ?(\S+).*?time ([0-9.]+) ms(?:  throughput ([0-9.]+))?', line)
        if m:
            key = m.group(1).replace('SpMM-', '').replace(', ', '_').replace(' ', '_')
            res[key] = dict(check=m.group(2), ms=float(m.group(3)), gflops=float(m.group(4)) if m.group(4) else None)
        elif line.startswith(('plan:', 'matrix ', 'hub-chain self-test', 'longest row')):
            res.setdefault('notes', []).append(line.strip())
    if out.returncode:
        res['stderr'] = out.stderr[-400:]
        res['stdout_tail'] = out.stdout[-400:]
    return res


def hipsparse_ms(rowptr, col, shape, N):
    dev = 'cuda'
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    nnz = col.shape[0]
    val = (torch.randint(0, 3, (nnz,), generator=g, device=dev).float() / 10)
    M = rowptr.shape[0] - 1
    A = torch.sparse_csr_tensor(torch.as_tensor(rowptr, device=dev).long(), torch.as_tensor(col, device=dev).long(), val,
                                size=(M, shape[1]))
    B = (torch.randint(0, 3, (A.shape[1], N), generator=g, device=dev).float() / 10)
    for _ in range(10):
        torch.sparse.mm(A, B)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        torch.sparse.mm(A, B)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 100


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r05_mtx'))
    ap.add_argument('--feats', default='32,64,128')
    a = ap.parse_args()
    feats = [int(x) for x in a.feats.split(',')]
    tmp = tempfile.mkdtemp(prefix='dgs_mtx_')
    for name, npz in MATRICES.items():
        z = np.load(npz)
        rowptr, col = z['rowptr'], z['col']
        shape = tuple(int(x) for x in z['shape'])
        mtx = os.path.join(tmp, name + '.mtx')
        write_mtx(mtx, rowptr, col, shape)
        rec = dict(matrix=name, rows=shape[0], cols=shape[1], nnz=int(col.shape[0]), max_degree=int(np.diff(rowptr).max()),
                   protocol='10 warm-up + 100 timed launches between events, values / dense in {0, .1, .2}, GFLOP/s = 2 nnz N / t '
                            '(reference example/ge-spmm/spmm.cu:145-215)', device=torch.cuda.get_device_name(0), feats={})
        for N in feats:
            d = run_driver(mtx, N)
            hs = hipsparse_ms(rowptr, col, shape, N)
            d['hipsparse_torch_sparse_mm'] = dict(ms=round(hs, 6), gflops=round(2.0 * col.shape[0] * N / hs * 1e-6, 2))
            rec['feats'][str(N)] = d
            print(name, N, json.dumps(d), flush=True)
        if name == 'p2p-Gnutella31':
            rec['published_v100_N32'] = PUBLISHED_V100
        with open(f'{a.out}_{name.replace("-", "_").lower()}.json', 'w') as f:
            json.dump(rec, f, indent=1)


if __name__ == '__main__':
    main()
