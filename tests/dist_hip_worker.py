"""One rank of the multi-GPU HIP parity check (launched by tests/test_gpu_dist.py under torch.distributed.run, backend
nccl = RCCL, one rank per GPU; also runnable by hand:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tests/dist_hip_worker.py).
Every rank builds the same seeded global graph, keeps its row block, runs DistSpMM with the HIP kernels (plain and
overlapped engines) and checks ITS rows of C / E / the gradients against the single-process CPU oracle on the whole
graph: the same assertions tests/test_dist_cpu.py makes on gloo with the oracle as compute stand-in."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dgsparse-lib_amd'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def main():
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    backend = os.environ.get('DGS_TEST_BACKEND', 'nccl')
    # backend gloo: every rank on cuda:0, collectives staged through the host (dgsparse.dist._a2a) - the N > 1 code paths with
    # the real HIP kernels on a box that has one GPU
    dev = torch.device('cuda', 0 if backend == 'gloo' else int(os.environ.get('LOCAL_RANK', '0')))
    torch.cuda.set_device(dev)
    if backend == 'gloo':
        dist.init_process_group('gloo')
    else:
        dist.init_process_group('nccl', device_id=dev)
    import oracle
    from bench import graphgen
    from dgsparse import dist as dd
    from util import assert_bitexact, assert_close, assert_sum_parity

    per = int(os.environ.get('DGS_TEST_ROWS_PER_RANK', '20000'))  # > 2^16 rows overall from 4 ranks on: planned schedule
    M, N = per * world, 32
    rp, col, st = graphgen.powerlaw_csr(M, 14 * M, alpha=2.0, dmax=M // 3, cols='powerlaw', seed=5)
    val = graphgen.weights(col.shape[0], 'tied', 5)
    X = (np.random.default_rng(1).integers(-2, 3, (M, N)) / 4).astype(np.float32)
    G = (np.random.default_rng(2).integers(-2, 3, (M, N)) / 4).astype(np.float32)
    part = dd.partition_csr(rp, col, val, world)[rank]
    part.rowptr, part.col, part.val = part.rowptr.to(dev), part.col.to(dev), part.val.to(dev)
    r0, r1 = part.row_offsets[rank], part.row_offsets[rank + 1]
    s0, s1 = int(rp[r0]), int(rp[r1])
    lens = np.diff(rp)
    C64 = oracle.spmm_sum_f64(rp, col, val, X)
    S64 = oracle.spmm_sum_f64(rp, col, val, X, absval=True)
    cp, rw, tv, _ = oracle.csr2csc(rp, col, val, M)
    Xl = torch.from_numpy(X[r0:r1].copy()).to(dev)
    Gl = torch.from_numpy(G[r0:r1].copy()).to(dev)
    for overlap in (False, True):
        eng = dd.DistSpMM(part, N, overlap=overlap)
        for red in ('sum', 'mean', 'max', 'min'):
            C = eng.spmm(Xl, red)
            Cg, Eg = oracle.spmm(red, rp, col, val, X, fma=True)
            tag = f'rank {rank}/{world} overlap={overlap} {red}'
            if red in ('max', 'min'):
                assert_bitexact(C.cpu().numpy(), Cg[r0:r1], tag + ' values')
                assert_bitexact(eng.last_E.cpu().numpy(), Eg[r0:r1], tag + ' E (global column ids)')
            else:
                sc = 1 if red == 'sum' else np.maximum(lens[r0:r1], 1)[:, None]
                assert_sum_parity(C.cpu().numpy(), Cg[r0:r1], C64[r0:r1] / sc, S64[r0:r1] / sc, 1e-5, 2e-6, tag,
                                  lens=lens[r0:r1])
            # both gradients through the reversed exchange
            Bl = Xl.clone().requires_grad_()
            vl = part.val.clone().requires_grad_()
            dd.DistSpMMFn.apply(eng, Bl, vl, red).backward(Gl)
            if red in ('sum', 'mean'):
                Gs = G if red == 'sum' else (G / np.maximum(lens, 1)[:, None]).astype(np.float32)
                gB, _ = oracle.spmm('sum', cp, rw, tv, Gs, fma=True)
                gW = oracle.sddmm(rp, col, Gs, X, fma=True)
            else:
                gB = oracle.spmm_mask(cp, rw, tv, G, Eg, fma=True)
                gW = oracle.sddmm_mask(rp, col, G, X, Eg, fma=True)
            if red in ('sum', 'mean'):
                # A^T G over SIGNED data: every schedule (one pass, or halo rows first + local rows) folds a hub column's
                # thousands of terms in its own order, so the bar is the sum bar (condition scale), as for the forward
                assert_sum_parity(Bl.grad.cpu().numpy(), gB[r0:r1], oracle.spmm_sum_f64(cp, rw, tv, Gs)[r0:r1],
                                  oracle.spmm_sum_f64(cp, rw, tv, Gs, absval=True)[r0:r1], 2e-5, 1e-5, tag + ' dB',
                                  lens=np.diff(cp)[r0:r1])
            else:
                assert_close(Bl.grad.cpu().numpy(), gB[r0:r1], 2e-5, 1e-5, tag + ' dB')
            assert_close(vl.grad.cpu().numpy(), gW[s0:s1], 2e-5, 1e-5, tag + ' dW')
        if overlap:
            # overlapped min in its corners (csrc/dist_merge.hip): signed zeros - MIN keeps the LATER operand's bits on a tie,
            # E the first arg - and NaN / inf features, where the merge has to give way to the sequential redo
            rng = np.random.default_rng(11)
            Xz = rng.choice(np.array([-0.0, 0.0, 0.0, 1.0, -1.0], np.float32), size=(M, N))
            Xn = X.copy()
            for bad in (np.nan, np.inf, -np.inf):
                Xn[rng.integers(0, M, 60), rng.integers(0, N, 60)] = bad
            for name, Xc, has_val in (('signed zeros', Xz, True), ('signed zeros, no values', Xz, False),
                                      ('NaN / inf features', Xn, True)):
                vc = val if has_val else None
                pc = dd.partition_csr(rp, col, vc, world)[rank]
                pc.rowptr, pc.col = pc.rowptr.to(dev), pc.col.to(dev)
                pc.val = None if pc.val is None else pc.val.to(dev)
                Cg, Eg = oracle.spmm('min', rp, col, vc, Xc, fma=True)
                for form in ('around', 'two'):  # ONE accumulating launch after the exchange (round 5, the default) / two
                    ec = dd.DistSpMM(pc, N, overlap=True, min_form=form)
                    assert ec.plan.rows_sorted and ec.min_form == form
                    Cc = ec.spmm(torch.from_numpy(Xc[r0:r1].copy()).to(dev), 'min')
                    assert_bitexact(Cc.cpu().numpy(), Cg[r0:r1], f'rank {rank}/{world} overlapped min ({form}), {name}: values')
                    assert_bitexact(ec.last_E.cpu().numpy(), Eg[r0:r1], f'rank {rank}/{world} overlapped min ({form}), {name}: E')
        remote = np.unique(col[s0:s1][(col[s0:s1] < r0) | (col[s0:s1] >= r1)])
        assert eng.n_halo == remote.shape[0] and eng.global_nnz == col.shape[0]
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    print(f'dist hip worker rank {rank}/{world} ok')


if __name__ == '__main__':
    main()
