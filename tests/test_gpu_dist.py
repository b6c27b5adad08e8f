"""Multi-GPU path on real hardware.

* ``test_c5_shard_*``: ONE rank's shard of BASELINE.json's 8-GPU configuration (C5: 16M x 16M, ~32 nnz/row, feat 64:
  2^21 rows and ~2^26 nnz per rank, relabelled into the [local | halo] operand) at full size on one GPU, through
  ``DistSpMM`` with the HIP kernels (standalone plan: the halo rows are filled in directly instead of exchanged), checked
  by size-independent properties + the oracle on sampled rows.
* ``test_dist_hip_multi_rank``: a torch.distributed.run-spawned RCCL job over min(device_count, 8) GPUs asserting each
  rank's C / E / gradients against the single-process oracle (skipped below 2 GPUs); the same job with ONE rank always
  runs, so process-group init, the collectives and the whole code path are exercised on a 1-GPU box too.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle
from util import assert_bitexact, assert_sum_parity

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_workers(world, extra_env=None):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', **(extra_env or {}))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'tests', 'dist_hip_worker.py')]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert r.stdout.count('ok') >= world, r.stdout[-2000:]


def test_dist_hip_single_rank_under_launcher():
    _run_workers(1, dict(DGS_TEST_ROWS_PER_RANK='70000'))  # > 2^16 rows: the planned row-stream schedule


@pytest.mark.parametrize('world,rows', [(2, '40000'), (3, '25000'), (4, '20000')])
def test_dist_hip_several_ranks_on_one_gpu(world, rows):
    """N > 1 with the REAL HIP kernels on a single-GPU box: every rank runs on cuda:0 under a gloo group and the
    collectives are staged through the host (dgsparse.dist._a2a).  Everything except the RCCL transport is the production
    code: halo plan, pack kernel, overlapped local / accumulating halo products (sum, mean, sorted max), E relabel to global
    ids, forward of every reduce and both gradients through the (overlapped) reversed exchange, each rank's rows checked
    against the single-process oracle on the whole graph."""
    _run_workers(world, dict(DGS_TEST_BACKEND='gloo', DGS_TEST_ROWS_PER_RANK=rows))


def test_bench_multi_rank_branch_dry_run_on_one_gpu():
    """bench.py --gpus 2 as the driver launches it, but with both ranks on cuda:0 over gloo (DGS_BENCH_BACKEND=gloo): the
    N > 1 branch - partition generator, halo plan across ranks, overlapped step, max-over-ranks timing, imbalance figures,
    exchange-only time, worst-case second run - executes end to end.  A functional check only: its numbers mean nothing."""
    import json
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', DGS_BENCH_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--rows-log2', '17', '--steps', '3',
           '--warmup', '1', '--settle', '2']
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert j['n_gpus'] == 2 and j['value'] > 0 and j['config']['parallelism'].startswith('rowpart2')
    assert j['halo']['rows_per_gpu'] > 0 and j['imbalance']['nnz']['max_over_mean'] >= 1.0 and j['exchange_only_ms'] > 0
    assert 'error' not in j.get('worst_case', {'error': 1}), j.get('worst_case')


def test_dist_hip_multi_rank():
    n = min(torch.cuda.device_count(), 8)
    if n < 2:
        pytest.skip('needs at least 2 GPUs (the driver\'s multi-GPU box runs it)')
    _run_workers(n)


@pytest.fixture(scope='module')
def c5_shard():
    """Rank 3 of 8 of the C5 graph: 2^21 rows, ~32 nnz/row, global columns in [0, 2^24)."""
    from dgsparse import dist as dd
    part = dd.synthetic_partition(3, 8, 1 << 21, 32, cols='powerlaw', locality=0.8, seed=0, device='cuda')
    eng = dd.DistSpMM(part, 64, standalone=True)
    return part, eng


def test_c5_shard_shape_and_properties(c5_shard):
    part, eng = c5_shard
    M, N = part.n_local, 64
    assert M == 1 << 21 and part.nnz > 0.9 * (1 << 26) and eng.n_halo > 0
    assert int(part.col.max()) < 8 * M and eng.plan.col_ext.max() < M + eng.n_halo
    deg = (part.rowptr[1:] - part.rowptr[:-1])
    # degree property (exact): all-ones features, unit weights
    eng.B_ext.fill_(1.0)
    C = eng.compute('sum', val=torch.ones_like(part.val))
    assert torch.equal(C[:, 0], deg.float()) and torch.equal(C[:, N - 1], deg.float())
    del C
    # column-id property: features = GLOBAL id of the row behind every [local | halo] slot -> max = largest global column
    # of the row (ids < 2^24 are exact in fp32) and E, mapped back from the extended space, names it
    gid = eng.plan.ext2glob.float()
    eng.B_ext.copy_(gid[:, None].expand(-1, N))
    C = eng.compute('max', val=torch.ones_like(part.val))
    nz = deg > 0
    seg_max = torch.zeros(M, dtype=torch.int64, device='cuda')
    rows = torch.repeat_interleave(torch.arange(M, device='cuda'), deg.long())
    seg_max.scatter_reduce_(0, rows, part.col.long(), 'amax', include_self=True)
    assert torch.equal(C[nz, 0], seg_max[nz].float()) and torch.equal(eng.last_E[nz, N - 1].long(), seg_max[nz])
    assert bool((eng.last_E[~nz] == -1).all())


@pytest.mark.parametrize('reduce', ['sum', 'max'])
def test_c5_shard_sampled_rows_vs_oracle(c5_shard, reduce):
    part, eng = c5_shard
    M, N = part.n_local, 64
    g = torch.Generator(device='cuda')
    g.manual_seed(7)
    eng.B_ext.copy_(torch.rand(eng.B_ext.shape, generator=g, device='cuda'))
    C = eng.compute(reduce)
    deg = (part.rowptr[1:] - part.rowptr[:-1])
    rows = torch.unique(torch.cat([torch.topk(deg, 8).indices, torch.randint(0, M, (300,), generator=g, device='cuda')]))
    rpc = part.rowptr.cpu().numpy()
    rows_c = rows.cpu().numpy()
    idx = np.concatenate([np.arange(rpc[r], rpc[r + 1]) for r in rows_c])
    srp = np.concatenate([[0], np.cumsum([rpc[r + 1] - rpc[r] for r in rows_c])]).astype(np.int32)
    t = torch.from_numpy(idx).cuda()
    scol = eng.plan.col_ext[t].cpu().numpy()       # the sub-CSR lives in the extended column space ...
    sval = part.val[t].cpu().numpy()
    Bh = eng.B_ext.cpu().numpy()                   # ... and reads the very buffer the kernel read
    Co, Eo = oracle.spmm(reduce, srp, scol, sval, Bh, fma=True)
    got = C[rows.long()].cpu().numpy()
    if reduce == 'max':
        assert_bitexact(got, Co, 'C5 shard max values')
        e2g = eng.plan.ext2glob.cpu().numpy()
        Eg = np.where(Eo >= 0, e2g[np.maximum(Eo, 0)], -1).astype(np.int32)
        assert_bitexact(eng.last_E[rows.long()].cpu().numpy(), Eg, 'C5 shard max E (global ids)')
    else:
        C64 = oracle.spmm_sum_f64(srp, scol, sval, Bh)
        assert_sum_parity(got, Co, C64, None, 1e-5, 2e-6, 'C5 shard sum', lens=np.diff(srp))


def test_bench_partitioned_path_under_the_launcher():
    """bench.py's multi-GPU branch (partition generator -> DistSpMM -> all-to-all-v -> timing -> worst-case second run)
    with ONE rank under torch.distributed.run: the code the driver's 2/4/8-GPU runs execute must not be dead code on the
    box that has a single GPU.  (With N > 1 from a bare shell bench.py launches these ranks itself.)"""
    import json
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--force-dist', '--rows-log2', '17',
           '--steps', '3', '--warmup', '1', '--no-cpu-baseline']
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert j['n_gpus'] == 1 and j['value'] > 0 and j['config']['parallelism'].startswith('rowpart1')
    assert 'halo' in j and 'error' not in j.get('worst_case', {'error': 1}), j.get('worst_case')
    assert j['roofline']['frac'] > 0 and j['scaling'] == 'weak'


# ('around': round 5, never run on hardware when written - collected last, so that a failure there cannot hide the forms that
# round 3 ran green: tests/conftest.py, marker first_contact)
@pytest.mark.parametrize('forms', ['merge+two', pytest.param('around', marks=pytest.mark.first_contact)])
@pytest.mark.parametrize('M,N,lo,hi,has_val', [(3000, 32, 900, 2100, True), (3000, 7, 900, 2100, True),
                                               (3000, 16, 0, 1500, True), (3000, 16, 1500, 3000, False),
                                               (3000, 8, 1200, 1200, True), (3000, 64, 0, 3000, True),
                                               (70000, 32, 20000, 50000, True), (70000, 12, 30000, 45000, False)])
def test_min_merge_and_nonfinite_flag_through_the_c_abi(M, N, lo, hi, has_val, forms):
    """dgs_spmm_csr_acc_min_f32 / _acc_min_around_f32 / dgs_spmm_min_merge_f32 / dgs_nonfinite_flag_f32 on their own: a matrix whose columns are
    cut in three ranges [lower | local | higher] the way a shard of dgsparse.dist is, every row's min put together from
    the three products - (a) lower folded in front of and higher behind the local result by the accumulating kernels, (b)
    the two halo halves computed apart and folded by the merge kernel - and compared bit for bit with the one-pass kernel
    AND the oracle: with signed zeros (the tie corner), with the flag forced up (the sequential redo), and with a NaN the
    detector has to find at any alignment.  The 70 000-row cases are past the single-launch size, so the accumulating
    commit runs in the row blocks, the unit blocks and the combine kernel."""
    from dgsparse import _capi
    from bench import graphgen
    dev = torch.device('cuda:0')
    K = M
    rp, col, st = graphgen.powerlaw_csr(M, 40 * M, K=K, alpha=2.0, dmax=max(900, M // 8), cols='powerlaw', seed=9)
    rng = np.random.default_rng(3)
    val = rng.choice(np.array([-1.0, 0.5, 1.0, 2.0], np.float32), size=col.shape[0])
    X = rng.choice(np.array([-0.0, 0.0, 0.0, 1.0, -1.0, 0.25], np.float32), size=(K, N))
    # local columns [lo, hi); ext ids: local -> c - lo, lower -> nl + c, higher -> nl + lo + (c - hi).  The parameters also
    # cover: no lower ranks, no higher ranks, no local columns at all, nothing remote (R = 0), no edge values
    if not has_val:
        val = None
    nl = hi - lo
    ext = np.where((col >= lo) & (col < hi), col - lo, np.where(col < lo, nl + col, nl + lo + col - hi)).astype(np.int32)
    Bext = np.concatenate([X[lo:hi], X[:lo], X[hi:]])
    rows = np.repeat(np.arange(M), np.diff(rp))
    is_loc = ext < nl

    def sub(mask, shift):
        cnt = np.bincount(rows[mask], minlength=M)
        r = np.zeros(M + 1, np.int32)
        r[1:] = np.cumsum(cnt)
        return r, (ext[mask] - shift).astype(np.int32), (None if val is None else val[mask])
    lrp, lcol, lval = sub(is_loc, 0)
    cnt_rem = np.bincount(rows[~is_loc], minlength=M)
    rem_rows = np.nonzero(cnt_rem)[0].astype(np.int32)
    cnt_lo = np.bincount(rows[~is_loc & (ext - nl < lo)], minlength=M)
    R = rem_rows.shape[0]
    rp2 = np.zeros(2 * R + 1, np.int64)
    rp2[1::2] = cnt_lo[rem_rows]
    rp2[2::2] = cnt_rem[rem_rows] - cnt_lo[rem_rows]
    rp2 = np.cumsum(rp2).astype(np.int32)
    hcol, hval = (ext[~is_loc] - nl).astype(np.int32), (None if val is None else val[~is_loc])
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    Bd = t(Bext)
    full_rp, full_col, full_val = t(rp), t(ext), t(val)
    Cref, Eref = oracle.spmm('min', rp, ext, val, Bext, fma=True)
    C1, E1 = _capi.spmm(_capi.MIN, full_rp, full_col, full_val, Bd)
    assert_bitexact(C1.cpu().numpy(), Cref, 'one-pass min')
    for force in ((0, 1) if forms == 'merge+two' else ()):
        flag = torch.full((1,), force, dtype=torch.int32, device=dev)
        if nl > 0:
            C, E = _capi.spmm(_capi.MIN, t(lrp), t(lcol), t(lval), Bd[:nl])
        else:  # a shard that owns no column: what an empty product leaves behind
            C, E = torch.zeros((M, N), device=dev), torch.full((M, N), -1, dtype=torch.int32, device=dev)
        if R == 0:
            assert_bitexact(C.cpu().numpy(), Cref, 'nothing remote: the local product is the result')
            continue
        Ch, Eh = _capi.spmm(_capi.MIN, t(rp2), t(hcol), t(hval), Bd[nl:])
        _capi.spmm_min_merge(t(rem_rows), t(rp2), Ch, Eh, nl, t(lrp), C, E, flag, full_rp, full_col, full_val, Bd)
        assert_bitexact(C.cpu().numpy(), Cref, f'merged min values (flag {force})')
        assert_bitexact(E.cpu().numpy(), Eref, f'merged min E (flag {force})')
    # (a) the schedule dgsparse.dist uses: two accumulating launches + the redo-only call
    is_lo = ~is_loc & (ext - nl < lo)

    def compact(mask):
        cnt = np.bincount(rows[mask], minlength=M)
        keep = np.nonzero(cnt)[0].astype(np.int32)
        r = np.zeros(keep.shape[0] + 1, np.int32)
        r[1:] = np.cumsum(cnt[keep])
        return r, (ext[mask] - nl).astype(np.int32), (None if val is None else val[mask]), keep
    for force in ((0, 1) if forms == 'merge+two' else ()):
        flag = torch.full((1,), force, dtype=torch.int32, device=dev)
        if nl > 0:
            C, E = _capi.spmm(_capi.MIN, t(lrp), t(lcol), t(lval), Bd[:nl])
        else:
            C, E = torch.zeros((M, N), device=dev), torch.full((M, N), -1, dtype=torch.int32, device=dev)
        for mask, first in ((is_lo, True), (~is_loc & ~is_lo, False)):
            srp, scol, sval, keep = compact(mask)
            if keep.shape[0]:
                srp_d, scol_d = t(srp), t(scol)
                # second round of the big cases: through the cached locality plan (None where the library offers none)
                plan = _capi.spmm_plan(srp_d, scol_d, K - nl, N) if (M > 3000 and force) else None
                _capi.spmm_acc_min(srp_d, scol_d, t(sval), Bd[nl:], C, E, t(keep), nl, first, plan=plan)
        if R:
            _capi.spmm_min_merge(t(rem_rows), None, None, None, 0, None, C, E, flag, full_rp, full_col, full_val, Bd)
        assert_bitexact(C.cpu().numpy(), Cref, f'accumulated min values (flag {force})')
        assert_bitexact(E.cpu().numpy(), Eref, f'accumulated min E (flag {force})')
    # (c) round 5, dgsparse.dist's default: ONE accumulating launch, the local result a virtual entry of its row
    # (dgs_spmm_csr_acc_min_around_f32; ids: lower slots | lo + shard row | higher slots + M), plan-free and over a plan
    if R and forms == 'around':
        hr, hs = rows[~is_loc], ext[~is_loc] - nl  # halo entries: row, slot (lower ranks' slots are [0, lo))
        has_loc = np.diff(lrp)[rem_rows] > 0
        vr = rem_rows[has_loc].astype(np.int64)
        ar = np.concatenate([hr, vr])
        ac = np.concatenate([np.where(hs < lo, hs, hs + M), lo + vr])
        order = np.lexsort((ac, ar))
        arp = np.zeros(R + 1, np.int32)
        arp[1:] = np.cumsum(np.bincount(ar, minlength=M)[rem_rows])
        acol = ac[order].astype(np.int32)
        aval = None if val is None else np.concatenate([val[~is_loc], np.ones(vr.shape[0], np.float32)])[order]
        arp_d, acol_d = t(arp), t(acol)
        for force in (0, 1):
            flag = torch.full((1,), force, dtype=torch.int32, device=dev)
            if nl > 0:
                C, E = _capi.spmm(_capi.MIN, t(lrp), t(lcol), t(lval), Bd[:nl])
            else:
                C, E = torch.zeros((M, N), device=dev), torch.full((M, N), -1, dtype=torch.int32, device=dev)
            plan = _capi.spmm_plan(arp_d, acol_d, K - nl + M, N) if (M > 3000 and force) else None
            _capi.spmm_acc_min_around(arp_d, acol_d, t(aval), Bd[nl:], C, E, t(rem_rows), nl, lo, M, plan=plan)
            _capi.spmm_min_merge(t(rem_rows), None, None, None, 0, None, C, E, flag, full_rp, full_col, full_val, Bd)
            assert_bitexact(C.cpu().numpy(), Cref, f'around-form min values (flag {force})')
            assert_bitexact(E.cpu().numpy(), Eref, f'around-form min E (flag {force})')
    # the detector: clean data leaves the flag alone; one NaN / inf anywhere (any alignment, head, tail) raises it
    for off in ((0, 1, 3) if forms == 'merge+two' else ()):
        x = torch.rand(100003, device=dev)[off:].contiguous() if off == 0 else torch.rand(100003 + off, device=dev)[off:]
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        _capi.nonfinite_flag(x.contiguous() if off == 0 else x, flag)
        assert int(flag) == 0
        for pos, bad in ((0, float('nan')), (x.numel() - 1, float('inf')), (x.numel() // 2 + 1, float('-inf'))):
            y = x.clone()
            if off:
                y = torch.empty(x.numel() + off, device=dev)[off:]
                y.copy_(x)
            y[pos] = bad
            flag.zero_()
            _capi.nonfinite_flag(y, flag)
            assert int(flag) == 1, (off, pos, bad)
