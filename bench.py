#!/usr/bin/env python3
"""bench.py -- headline benchmark of the CSR SpMM hot path on MI355X.

Metric (BASELINE.json): CSR SpMM GFLOP/s (+ achieved HBM GB/s vs the 8 TB/s roofline), feat=64, on a synthetic
power-law CSR with 2^20 rows and ~16 nnz/row PER GPU (north_star's 1Mx1M graph; at N>1 the graph is
(N*2^20)x(N*2^20), 1-D row-partitioned, halo feature rows exchanged by RCCL all-to-all-v => weak scaling).
FLOP convention 2*nnz*N (reference example/ge-spmm/spmm.cu:162).  One "step" = one SpMM-sum over the whole
graph with inputs resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--feat 64] [--reduce sum] [--cols powerlaw|uniform]

For N>1 launch under torch.distributed.run (one rank per GPU); rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'dgsparse-lib_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import graphgen  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling


def alg_bytes_spmm(M, K, N, nnz, has_value=True, with_E=False):
    """every array touched exactly once."""
    return 4 * (M + 1) + 4 * nnz * (2 if has_value else 1) + 4 * K * N + 4 * M * N * (2 if with_E else 1)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--feat', type=int, default=64)
    ap.add_argument('--reduce', default='sum', choices=['sum', 'mean', 'max', 'min'])
    ap.add_argument('--cols', default='powerlaw', choices=['powerlaw', 'uniform', 'local'])
    ap.add_argument('--rows-log2', type=int, default=20, help='rows per GPU = 2^k')
    ap.add_argument('--deg', type=int, default=16)
    ap.add_argument('--locality', type=float, default=0.8,
                    help='N>1: probability that an edge stays inside its row partition (1-edge-cut)')
    ap.add_argument('--alpha', type=float, default=2.1, help='power-law exponent of the degree law')
    ap.add_argument('--dmax', type=int, default=1 << 16, help='degree cap')
    ap.add_argument('--ncols', type=int, default=0, help='columns of A / rows of the dense operand (0 = square)')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--force-dist', action='store_true', help='run the partitioned code path even with one rank (testing)')
    ap.add_argument('--sweep', action='store_true', help='also time feat=32/128 and max (extra keys)')
    ap.add_argument('--no-dense', action='store_true', help='skip the Reddit-shaped side measurement (extra key)')
    return ap.parse_args()


def time_steps(fn, steps, warmup, dist_on):
    """W untimed warm-ups, then exactly K steps between barrier+synchronize; also HIP-event time on the stream."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return wall, ev0.elapsed_time(ev1) / 1e3


def cpu_baseline(rp, col, val, X, flops):
    """The reference's own single-threaded CPU loop (oracle/_ref: spmm_reference_host, sp_util.hpp:63-84) when it
    was built, else the C restatement; plus the OpenMP restatement on all host cores.  Bounded: one pass each."""
    import oracle
    out = {}
    M = rp.shape[0] - 1
    # bounded sample: the first `rows` rows such that the work is <= ~2^24 nnz (the whole 1M-row graph)
    kind = 'reference' if oracle.have_ref() else 'port'
    fn = (lambda: oracle.ref_spmm_sum(rp, col, val, X)) if kind == 'reference' else \
         (lambda: oracle.spmm('sum', rp, col, val, X, threads=1))
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter() - t0
    out['cpu_baseline'] = dict(value=round(flops / t1 / 1e9, 3), unit='GFLOP/s', cores=1, kind=kind,
                               sample=f'whole workload, 1 pass ({M} rows, {col.shape[0]} nnz, N={X.shape[1]}), {t1:.2f} s')
    nthr = min(os.cpu_count() or 1, oracle.max_threads())
    best = 1e30
    for _ in range(2):
        t0 = time.perf_counter()
        oracle.spmm('sum', rp, col, val, X, threads=nthr)
        best = min(best, time.perf_counter() - t0)
    out['cpu_baseline_all_cores'] = dict(value=round(flops / best / 1e9, 3), unit='GFLOP/s', cores=nthr, kind='port',
                                         sample=f'whole workload, best of 2 passes, OpenMP over rows, {best:.2f} s')
    return out


def main():
    a = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist_on = world > 1
    if a.gpus != world and dist_on:
        raise SystemExit(f'--gpus {a.gpus} but WORLD_SIZE={world}')
    if a.gpus > 1 and not dist_on:
        raise SystemExit('for --gpus N>1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py ...')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if dist_on:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.distributed.init_process_group('nccl', device_id=dev)

    import dgsparse
    from dgsparse import _capi

    Mloc = 1 << a.rows_log2
    N = a.feat
    op = {'sum': _capi.SUM, 'mean': _capi.MEAN, 'max': _capi.MAX, 'min': _capi.MIN}[a.reduce]
    extra = {}

    use_dist = dist_on or a.force_dist
    if not use_dist:
        rp, col, st = graphgen.powerlaw_csr(Mloc, Mloc * a.deg, K=(a.ncols or None), alpha=a.alpha, dmax=a.dmax, cols=a.cols, seed=a.seed,
                                            device=str(dev), as_torch=True)
        K = st['K']
        g = torch.Generator(device=dev)
        g.manual_seed(a.seed + 1)
        val = torch.rand(st['nnz'], generator=g, device=dev)
        X = torch.rand((K, N), generator=g, device=dev)
        nnz_total = st['nnz']

        def step():
            return _capi.spmm(op, rp, col, val, X)

        # parity spot-check on the exact tensors being timed: 2048 sampled rows (always including the longest)
        # against an fp64 torch gather-sum; the full parity suite is tests/ -m gpu
        C, _ = step()
        if a.reduce in ('sum', 'mean'):
            deg = (rp[1:] - rp[:-1]).long()
            gsel = torch.Generator(device=dev)
            gsel.manual_seed(123)
            sel = torch.cat([torch.randint(0, Mloc, (2047,), generator=gsel, device=dev), deg.argmax().view(1)])
            worst = 0.0
            for r in sel.tolist()[-64:] + sel.tolist()[:256]:
                s0, e0 = int(rp[r]), int(rp[r + 1])
                if e0 == s0:
                    continue
                ref = (val[s0:e0].double()[:, None] * X[col[s0:e0].long()].double()).sum(0)
                if a.reduce == 'mean':
                    ref = ref / (e0 - s0)
                worst = max(worst, ((C[r].double() - ref).abs() / ref.abs().clamp_min(1e-6)).max().item())
            assert worst < 1e-5, f'bench self-check failed: rel err {worst}'
            extra['self_check_max_rel_err'] = worst
        del C
        b_alg = alg_bytes_spmm(Mloc, K, N, nnz_total, True, a.reduce in ('max', 'min'))
        parallelism = 'single'
        workload = f'synthetic power-law CSR {Mloc}x{K}, nnz={nnz_total} (~{nnz_total / Mloc:.1f}/row, alpha={a.alpha}, ' \
                   f'max_deg={st["max_deg"]}), cols={a.cols}, SpMM-{a.reduce} feat={N}, fp32 values'
    else:
        from dgsparse import dist as ddist
        part = ddist.synthetic_partition(rank, world, Mloc, a.deg, cols=a.cols, locality=a.locality, seed=a.seed,
                                         device=dev)
        g = torch.Generator(device=dev)
        g.manual_seed(a.seed + 1 + rank)
        eng = ddist.DistSpMM(part, N)
        Xloc = eng.local_features()  # features live inside the exchange buffer: no per-call copy
        Xloc.copy_(torch.rand((Mloc, N), generator=g, device=dev))
        nnz_total = eng.global_nnz

        def step():
            return eng.spmm(Xloc, a.reduce)

        step()
        K = Mloc * world
        b_alg = alg_bytes_spmm(Mloc, Mloc + eng.n_halo, N, part.nnz, True, a.reduce in ('max', 'min'))
        parallelism = f'rowpart{world}+halo-alltoallv'
        extra['halo'] = dict(rows_per_gpu=int(eng.n_halo), bytes_per_gpu=int(eng.n_halo) * N * 4, locality=a.locality)
        workload = f'synthetic power-law CSR {K}x{K} ({Mloc} rows/GPU, ~{a.deg}/row), cols={a.cols}, ' \
                   f'locality={a.locality}, SpMM-{a.reduce} feat={N}, 1-D row partition + halo all-to-all-v'

    wall, ev = time_steps(step, a.steps, a.warmup, dist_on)
    t = torch.tensor([wall, ev], device=dev, dtype=torch.float64)
    if dist_on:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    wall, ev = t.tolist()
    flops = 2.0 * nnz_total * N
    ms = wall / a.steps * 1e3
    kern_s = ev / a.steps  # average per-launch duration from HIP events on the launch stream
    achieved = b_alg / kern_s / 1e9

    res = {
        'metric': f'CSR SpMM GFLOP/s (feat={N}, {a.reduce})',
        'value': round(flops / (wall / a.steps) / 1e9, 2),
        'unit': 'GFLOP/s',
        'n_gpus': world,
        'steps': a.steps,
        'warmup': a.warmup,
        'ms_per_step': round(ms, 5),
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': workload, 'rows_per_gpu': Mloc, 'nnz_total': int(nnz_total), 'feat': N,
                   'reduce': a.reduce, 'cols': a.cols, 'seed': a.seed, 'parallelism': parallelism},
        'roofline': {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': None,
                     'alg_bytes_per_launch': int(b_alg), 'kernel_us': round(kern_s * 1e6, 2)},
    }
    res.update(extra)
    tf = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    if os.path.exists(tf) and not use_dist:
        try:
            tj = json.load(open(tf))
            key = f'{a.reduce}_feat{N}_{a.cols}'
            if key in tj:
                res['roofline']['traffic'] = tj[key]
        except Exception:
            pass

    if a.sweep and not use_dist and rank == 0:
        sw = {}
        for n2, red in ((32, 'sum'), (128, 'sum'), (64, 'max'), (64, 'mean')):
            X2 = torch.rand((K, n2), device=dev)
            o2 = {'sum': _capi.SUM, 'max': _capi.MAX, 'mean': _capi.MEAN}[red]
            w2, e2 = time_steps(lambda: _capi.spmm(o2, rp, col, val, X2), 20, 3, False)
            b2 = alg_bytes_spmm(Mloc, K, n2, nnz_total, True, red == 'max')
            sw[f'{red}_feat{n2}'] = dict(gflops=round(2.0 * nnz_total * n2 / (w2 / 20) / 1e9, 1),
                                         gbs=round(b2 / (e2 / 20) / 1e9, 1), frac=round(b2 / (e2 / 20) / 1e9 / HBM_PEAK_GBS, 4))
        res['sweep'] = sw

    if rank == 0 and not use_dist and not a.no_dense:
        # side figure (extra key, not the metric): the dense-graph configuration of BASELINE.json (C3), which takes the
        # column-panel schedule; ~10 s of graph generation on the GPU + 10 timed launches
        try:
            rp3, col3, st3 = graphgen.dataset_shaped('reddit', seed=0, device=str(dev), as_torch=True)
            val3 = torch.rand(st3['nnz'], device=dev)
            X3 = torch.rand((st3['K'], 128), device=dev)
            w3, e3 = time_steps(lambda: _capi.spmm(_capi.SUM, rp3, col3, val3, X3), 10, 3, False)
            b3 = alg_bytes_spmm(st3['M'], st3['K'], 128, st3['nnz'], True, False)
            res['dense_graph'] = dict(
                workload=f"Reddit-shaped {st3['M']}x{st3['K']}, nnz {st3['nnz']}, feat 128, sum",
                schedule=_capi.spmm_schedule(_capi.SUM, st3['M'], st3['K'], 128, st3['nnz']),
                ms_per_step=round(w3 / 10 * 1e3, 4), gflops=round(2.0 * st3['nnz'] * 128 / (w3 / 10) / 1e9, 1),
                alg_gbs=round(b3 / (e3 / 10) / 1e9, 1), frac=round(b3 / (e3 / 10) / 1e9 / HBM_PEAK_GBS, 4))
            del rp3, col3, val3, X3
        except Exception as e:
            res['dense_graph'] = dict(error=str(e))

    if rank == 0 and not use_dist and not a.no_cpu_baseline:
        try:
            res.update(cpu_baseline(rp.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy(), X.cpu().numpy(), flops))
        except Exception as e:  # the baseline is a reported side figure; never lose the GPU line over it
            res['cpu_baseline'] = dict(value=None, unit='GFLOP/s', cores=0, kind='port', sample=f'failed: {e}')
    if rank == 0:
        print(json.dumps(res))
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
