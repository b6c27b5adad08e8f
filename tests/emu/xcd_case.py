"""TEST INFRASTRUCTURE ONLY: one planned sum with DGS_HUB_XCD=1 (hub rows chained slice by slice across the XCDs) against the
oracle; run by tests/test_emu_cpu.py in a subprocess under DGS_EMU_BLOCKS / DGS_EMU_BLOCK_ORDER (read once per process)."""
import sys, time, numpy as np
import os
HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE), HERE):
    sys.path.insert(0, p)
import oracle, emu_lib as E
rng = np.random.default_rng(1)
M, K, N = 66000, 5000, 64
deg = rng.integers(0, 2, M)
for r, d in ((100, 3000), (7000, 1500), (65000, 1100), (50000, 1025), (9, 2500), (300, 200)):
    deg[r] = d
rp = np.zeros(M + 1, np.int32); rp[1:] = np.cumsum(deg)
col = np.concatenate([np.sort(rng.choice(K, d, replace=False)) for d in deg]).astype(np.int32)
val = rng.random(col.size, dtype=np.float32)
X = rng.random((K, N), dtype=np.float32)
ref, _ = oracle.spmm('sum', rp, col, val, X, fma=True)
hub = deg > 1024
E.set_env(DGS_HUB_CHAIN=1024, DGS_NBU=8, DGS_HUB_XCD=int(sys.argv[1]) if len(sys.argv) > 1 else 1)
plan = E.spmm_plan(rp, col, K)
t0 = time.time()
C, _ = E.spmm(E.SUM, rp, col, val, X, plan=plan)
print('hub mismatches', int((C[hub].view(np.int32) != ref[hub].view(np.int32)).sum()), 'all within 1e-5',
      bool((np.abs(C - ref) <= 1e-5 * np.abs(ref) + 1e-6).all()), f'{time.time() - t0:.1f} s')
