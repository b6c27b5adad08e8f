#!/usr/bin/env python3
"""CSR of the reference's second benchmark matrix, example/data/ca-CondMat.mtx (SNAP public dataset; a data file, like
p2p-Gnutella31 in make_golden.py), read with the reference loader's semantics (pattern only, symmetric entries mirrored,
sorted, deduplicated: example/util/sp_util.hpp:171-251 via dgsparse/io.py, which tests/test_host_cpu.py pins against the
reference's own read_mtx_file).  The reference's published benchmark (example/README.md:47-60) and its drivers
(example/ge-spmm/spmm.cu, example/sddmm/sddmm.cu) run on these two files; bench/mtx_bench.py and
tests/test_gpu_parity.py::test_real_graphs_all_reduces use the CSR arrays on the GPU box, where /root/reference is absent.

Usage: python tests/golden/make_mtx_golden.py   (needs /root/reference)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'dgsparse-lib_amd'))
from dgsparse import io as dio  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

if __name__ == '__main__':
    nrow, ncol, rowptr, col = dio.read_mtx('/root/reference/example/data/ca-CondMat.mtx')
    np.savez_compressed(os.path.join(OUT, 'ca_condmat_csr.npz'), rowptr=rowptr.astype(np.int32), col=col.astype(np.int32),
                        shape=np.array([nrow, ncol]))
    print('ca-CondMat', nrow, ncol, col.shape[0], 'nnz, max degree', int(np.diff(rowptr).max()),
          os.path.getsize(os.path.join(OUT, 'ca_condmat_csr.npz')) // 1024, 'KiB')
