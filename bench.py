#!/usr/bin/env python3
"""bench.py -- headline benchmark of the CSR SpMM hot path on MI355X.

Metric (BASELINE.json): CSR SpMM GFLOP/s (+ achieved HBM GB/s vs the 8 TB/s roofline), feat=64, on a synthetic
power-law CSR with 2^20 rows and ~16 nnz/row PER GPU (north_star's 1Mx1M graph; at N>1 the graph is
(N*2^20)x(N*2^20), 1-D row-partitioned, halo feature rows exchanged by RCCL all-to-all-v => weak scaling).
FLOP convention 2*nnz*N (reference example/ge-spmm/spmm.cu:162).  One "step" = one SpMM-sum over the whole
graph with inputs resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--feat 64] [--reduce sum] [--cols powerlaw|uniform]

``--gpus N`` with N > 1 from a bare shell re-launches itself under ``python -m torch.distributed.run`` (one rank per
GPU, rendezvous on 127.0.0.1); under an existing launcher (WORLD_SIZE set) it just joins.  Rank 0 prints ONE JSON line.

Protocol: W warm-ups + K timed steps between barrier+synchronize (the contract's number); beside it
the median of 5 repeats of the same K launches between HIP events on the launch stream, a unit-weight run, and seeds
1..4 of the same generator (extra keys; --no-protocol skips them).
Before the W warm-ups the step is launched --settle (default 50) more times, untimed and reported as settle_steps: an
idle MI355X needs ~25 launches before its step time settles (0.404 -> 0.394 ms on the default workload).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'dgsparse-lib_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

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
    ap.add_argument('--plan', type=int, default=-1, help='1/0: force the cached locality plan on/off (default: auto)')
    ap.add_argument('--strict', default='', choices=['', 'fma', 'nofma'],
                    help='time the strict-order schedule (DGS_ALG_STRICT_SUM / _NOFMA: every row one sequential chain, '
                         'bit-exact against the sequential reference) instead of the default one; without the flag the '
                         'strict schedule is still measured and checked beside the headline (key "strict")')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-protocol', action='store_true', help='skip median-of-5 / unit-weight / seeds 1..4 extras')
    ap.add_argument('--protocol-seeds', type=int, default=5, help='graph seeds 0 .. n-1 timed and parity-checked by the protocol leg '
                    '(SURVEY 8(d): five; the CPU dry run of tests/ walks the code with two)')
    ap.add_argument('--settle', type=int, default=50, help='untimed launches before the W warm-ups: an idle MI355X needs '
                    '~25 launches (10 ms) before the step time settles (0.404 -> 0.393 ms); reported as settle_steps')
    ap.add_argument('--no-worst-case', action='store_true', help='N>1: skip the uniform-columns, locality-0 second run')
    ap.add_argument('--force-dist', action='store_true', help='run the partitioned code path even with one rank (testing)')
    ap.add_argument('--sweep', action='store_true', help='also time feat=32/128 and max (extra keys)')
    ap.add_argument('--no-dense', action='store_true', help='skip the Reddit-shaped side measurement (extra key)')
    ap.add_argument('--fold-inprocess', action='store_true', help='run the in-kernel-fold leg inside this process (default: a child '
                    'process - the leg forces an opt-in schedule that has never met the hardware, and a GPU fault there must not '
                    'cost the line)')
    ap.add_argument('--only-fold-leg', action='store_true', help=argparse.SUPPRESS)
    return ap.parse_args()


def self_spawn(n):
    """`python bench.py --gpus N` from a bare shell: become N ranks under torch.distributed.run and relay its exit code."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    env.setdefault('NCCL_DEBUG', 'WARN')  # first contact with an 8-GPU node: keep what RCCL has to say (see nccl_log_tail)
    env.setdefault('NCCL_DEBUG_FILE', os.path.join('/tmp', 'dgs_bench_nccl_%h_%p.log'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def nccl_log_tail(limit=1500):
    """Tail of the RCCL debug files of this launch (NCCL_DEBUG_FILE pattern set by self_spawn / the launcher), for the JSON line
    of a failed multi-GPU run: the driver keeps stdout, not the ranks' files."""
    import glob
    pat = os.environ.get('NCCL_DEBUG_FILE', '')
    if not pat:
        return None
    out = []
    for f in sorted(glob.glob(pat.replace('%h', '*').replace('%p', '*')))[:16]:
        try:
            txt = open(f).read().strip()
        except OSError:
            continue
        if txt:
            out.append(os.path.basename(f) + ': ' + txt[-300:])
    return ' | '.join(out)[-limit:] if out else None


def time_steps(fn, steps, warmup, dist_on):
    """W untimed warm-ups, then exactly K steps between barrier+synchronize; also HIP-event time on the stream."""
    import torch
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


def event_ms(fn, steps):
    """Average per-launch time of `steps` launches between two HIP events on the launch stream (no barrier)."""
    import torch
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(steps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / steps


def cpu_baseline(rp, col, val, X, flops, C_gpu=None, C_strict=None):
    """The reference's own single-threaded CPU loop (oracle/_ref: spmm_reference_host, sp_util.hpp:63-84) when it
    was built, else the C restatement; plus the OpenMP restatement on all host cores.  Bounded: one pass each.
    The pass also yields the sequential fp32 result of EVERY row, so the GPU result of the timed tensors is compared
    against it here (the checker leg; never part of the timed region)."""
    import numpy as np
    import oracle
    out = {}
    M = rp.shape[0] - 1
    kind = 'reference' if oracle.have_ref() else 'port'
    fn = (lambda: oracle.ref_spmm_sum(rp, col, val, X)) if kind == 'reference' else \
         (lambda: oracle.spmm('sum', rp, col, val, X, threads=1)[0])
    t0 = time.perf_counter()
    Cseq = fn()
    t1 = time.perf_counter() - t0
    out['cpu_baseline'] = dict(value=round(flops / t1 / 1e9, 3), unit='GFLOP/s', cores=1, kind=kind,
                               sample=f'whole workload, 1 pass ({M} rows, {col.shape[0]} nnz, N={X.shape[1]}), {t1:.2f} s')
    if C_gpu is not None and Cseq is not None:
        Cseq = np.asarray(Cseq).reshape(C_gpu.shape)
        rel = np.abs(C_gpu.astype(np.float64) - Cseq) / np.maximum(np.abs(Cseq), 1e-6)
        lens = np.diff(rp)
        short = lens <= 64
        # rows of 10^3..10^4 nnz: the reference's own sequential fp32 chain drifts from the exact sum by more than 1e-5;
        # the float64 yardstick says which of the two results is off (the fixed-tree split sum is the closer one)
        C64 = oracle.spmm_sum_f64(rp, col, val, X)
        e_gpu = np.abs(C_gpu.astype(np.float64) - C64) / np.maximum(np.abs(C64), 1e-6)
        e_seq = np.abs(Cseq.astype(np.float64) - C64) / np.maximum(np.abs(C64), 1e-6)
        out['parity'] = dict(
            max_rel_err_vs_fp64_gpu=float(e_gpu.max()), max_rel_err_vs_fp64_sequential_reference=float(e_seq.max()),
            oracle=('reference spmm_reference_host (sequential fp32, no fma)' if kind == 'reference'
                    else 'oracle sequential fp32'), rows_checked=int(M),
            max_rel_err_vs_sequential=float(rel.max()),
            max_rel_err_rows_le_64nnz=float(rel[short].max()) if short.any() else 0.0,
            max_rel_err_rows_gt_64nnz=float(rel[~short].max()) if (~short).any() else 0.0,
            longest_row_nnz=int(lens.max()), within_1e_5=bool(rel.max() <= 1e-5))
        far = rel > 1e-5  # where the two fp32 results differ by more than the bar: which one is off?
        out['parity'].update(elements_beyond_1e_5=int(far.sum()), elements=int(rel.size),
                             gpu_closer_to_fp64_on_all_of_them=bool((e_gpu[far] <= e_seq[far]).all()) if far.any() else True)
    if C_strict and Cseq is not None:
        # the strict-order schedule against the same sequential results: NOFMA is the reference's host loop bit for bit
        # (g++ emits no fused multiply-add for it), FMA is the reference kernel's contraction (checked against the oracle's
        # fmaf chain)
        Cseq = np.asarray(Cseq).reshape(M, -1)
        ps = {}
        for mode, Cg in C_strict.items():
            seq = Cseq if mode == 'nofma' else oracle.spmm('sum', rp, col, val, X, fma=True, threads=min(os.cpu_count() or 1, oracle.max_threads()))[0]
            rel = np.abs(Cg.astype(np.float64) - Cseq) / np.maximum(np.abs(Cseq), 1e-6)
            ps[mode] = dict(bit_exact_vs_its_sequential_chain=bool(np.array_equal(Cg.view(np.int32), np.asarray(seq).view(np.int32))),
                            chain=('reference spmm_reference_host (mul, add)' if mode == 'nofma' and kind == 'reference' else
                                   'oracle sequential ' + ('fmaf' if mode == 'fma' else 'mul, add')),
                            max_rel_err_vs_sequential_reference=float(rel.max()), within_1e_5=bool(rel.max() <= 1e-5),
                            elements_beyond_1e_5=int((rel > 1e-5).sum()), rows_checked=int(M))
        out['parity_strict'] = ps
    nthr = min(os.cpu_count() or 1, oracle.max_threads())
    best = 1e30
    for _ in range(2):
        t0 = time.perf_counter()
        oracle.spmm('sum', rp, col, val, X, threads=nthr)
        best = min(best, time.perf_counter() - t0)
    out['cpu_baseline_all_cores'] = dict(value=round(flops / best / 1e9, 3), unit='GFLOP/s', cores=nthr, kind='port',
                                         sample=f'whole workload, best of 2 passes, OpenMP over rows, {best:.2f} s')
    return out


def parity_vs_sequential(C, rp, col, val, X):
    """Every element of a GPU sum against the sequential fp32 chain of algorithm 0 (checker leg, outside any timed region)."""
    import numpy as np
    import oracle
    Cseq = oracle.spmm('sum', rp, col, val, X, fma=False, threads=min(os.cpu_count() or 1, oracle.max_threads()))[0]
    rel = np.abs(C.astype(np.float64) - Cseq) / np.maximum(np.abs(Cseq), 1e-6)
    lens = np.diff(rp)
    return dict(max_rel_err_vs_sequential=float(rel.max()), within_1e_5=bool(rel.max() <= 1e-5),
                elements_beyond_1e_5=int((rel > 1e-5).sum()), elements=int(rel.size), longest_row_nnz=int(lens.max()),
                rows_gt_hub_threshold=int((lens > 16384).sum()))


def main():
    a = parse()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_spawn(a.gpus))

    import numpy as np
    import torch

    from bench import graphgen

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist_on = world > 1
    pg = dist_on or ('RANK' in os.environ and a.force_dist)  # process group (also for a 1-rank launcher run: tests)
    if a.gpus != world and dist_on:
        raise SystemExit(f'--gpus {a.gpus} but WORLD_SIZE={world}')
    # DGS_BENCH_BACKEND=gloo: functional dry run of the N > 1 branch on a box with ONE GPU - every rank on cuda:0, collectives
    # staged through the host by dgsparse.dist (tests/test_gpu_dist.py); the numbers of such a run mean nothing
    one_gpu = os.environ.get('DGS_BENCH_BACKEND') == 'gloo'
    if one_gpu:
        local_rank = 0
    if not torch.cuda.is_available() or local_rank >= torch.cuda.device_count():
        raise SystemExit(f'bench.py needs {max(a.gpus, 1)} GPU(s); visible: {torch.cuda.device_count()}')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if pg:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('NCCL_DEBUG', 'WARN')
        try:
            if one_gpu:
                torch.distributed.init_process_group('gloo')
            else:
                torch.distributed.init_process_group('nccl', device_id=dev)
            # first contact: one all-reduce of ones must count every rank before anything is timed
            ones = torch.ones(1, device=dev, dtype=torch.float32)
            if one_gpu:
                h = ones.cpu()
                torch.distributed.all_reduce(h)
                ones.copy_(h)
            else:
                torch.distributed.all_reduce(ones)
            torch.cuda.synchronize()
            rccl_ranks = int(round(ones.item()))
            if rccl_ranks != world:
                raise RuntimeError(f'all-reduce of ones counted {rccl_ranks} ranks, WORLD_SIZE is {world}')
        except Exception as e:  # noqa: BLE001  (a line the driver can read beats a traceback on 8 stderr streams)
            if rank == 0:
                print(json.dumps({'metric': f'CSR SpMM GFLOP/s (feat={a.feat}, {a.reduce})', 'value': None, 'unit': 'GFLOP/s',
                                  'n_gpus': world, 'error': f'process group / first collective failed: {e!r}',
                                  'backend': 'gloo' if one_gpu else 'nccl', 'nccl_log_tail': nccl_log_tail()}))
            raise

    def allreduce_max(x):  # gloo (dry run) reduces on the host
        if one_gpu:
            h = x.cpu()
            torch.distributed.all_reduce(h, op=torch.distributed.ReduceOp.MAX)
            x.copy_(h)
        else:
            torch.distributed.all_reduce(x, op=torch.distributed.ReduceOp.MAX)


    import dgsparse  # noqa: F401
    from dgsparse import _capi

    Mloc = 1 << a.rows_log2
    N = a.feat
    op = {'sum': _capi.SUM, 'mean': _capi.MEAN, 'max': _capi.MAX, 'min': _capi.MIN}[a.reduce]
    extra = {}
    if pg:
        extra['rccl_ranks'] = rccl_ranks
        extra['backend'] = 'gloo (one-GPU dry run)' if one_gpu else 'nccl (RCCL)'

    def make_graph(seed):
        # counter-based sampler (bench/graphgen.py hash_bits): the SAME graph, values and features on the GPU and on the CPU, so
        # the emulation (bench/emu_parity.py) and the tests check exactly the tensors timed here (VERDICT r4 #2b)
        rp_, col_, st_ = graphgen.powerlaw_csr(Mloc, Mloc * a.deg, K=(a.ncols or None), alpha=a.alpha, dmax=a.dmax,
                                               cols=a.cols, seed=seed, device=str(dev), as_torch=True, sampler='hash')
        val_ = graphgen.values_t(st_['nnz'], seed, dev)
        X_ = graphgen.features_t(st_['K'], N, seed, dev)
        return rp_, col_, st_, val_, X_

    strict_alg = {'': 0, 'fma': _capi.ALG_STRICT_SUM, 'nofma': _capi.ALG_STRICT_NOFMA}[a.strict]

    def make_step(rp_, col_, val_, X_):
        """One SpMM through the C ABI; with a plan (Storage-cached locality plan, built once outside the timed region)
        when the library offers one for this shape."""
        plan = None
        if a.plan != 0 and hasattr(_capi, 'spmm_plan'):
            plan = _capi.spmm_plan(rp_, col_, X_.shape[0], N, force=(a.plan == 1))
        if strict_alg:  # (over the plan's strict table when there is a plan: one launch, no classify pass)
            return (lambda: _capi.spmm(op, rp_, col_, val_, X_, algorithm=strict_alg, plan=plan)), plan is not None
        if plan is not None:
            return (lambda: _capi.spmm(op, rp_, col_, val_, X_, plan=plan)), True
        return (lambda: _capi.spmm(op, rp_, col_, val_, X_)), False

    use_dist = dist_on or a.force_dist
    C_check = None
    if not use_dist:
        rp, col, st, val, X = make_graph(a.seed)
        K = st['K']
        nnz_total = st['nnz']
        step, planned = make_step(rp, col, val, X)

        def fold_leg():
            """The in-kernel fold of partial rows (opt-in: DGS_FOLD=1 | 2) next to the combine launch it replaces, on the same
            tensors: first the device's own verdict on the hand-over (dgs_spmm_fold_selftest: every family of partial row, 3
            rounds, the last under a streaming load), then DGS_FOLD = 1 / 0 for one measurement each and a bit compare of the
            two results, then back to the process's own setting.  Decision rule: the fold becomes a default
            only with on_ms < off_ms on hardware."""
            had = os.environ.get('DGS_FOLD')
            try:
                verdict, fam = _capi.fold_selftest(rounds=3, load=True)
            except Exception as e:
                verdict, fam = -2, str(e)
            fd = dict(selftest=verdict, selftest_mismatches_per_family=fam, gate=_capi.fold_gate(),
                      default_on=bool(had == '1' or (had == '2' and _capi.fold_gate() > 0)))
            outs = {}
            for name, v in (('on_ms', '1'), ('off_ms', '0')):
                if v == '1' and verdict != 1:
                    fd[name] = None  # a hand-over the device got wrong is not timed
                    continue
                os.environ['DGS_FOLD'] = v
                _capi.reload_tuning()
                stepf, _ = make_step(rp, col, val, X)
                outs[v] = stepf()[0].clone()
                fd[name] = round(sorted(event_ms(stepf, max(10, a.steps // 5)) for _ in range(3))[1], 5)
                del stepf
            if len(outs) == 2:
                fd['same_bits'] = bool(torch.equal(outs['1'].view(torch.int32), outs['0'].view(torch.int32)))
            del outs
            if had is None:
                os.environ.pop('DGS_FOLD')
            else:
                os.environ['DGS_FOLD'] = had
            _capi.reload_tuning()
            fd['note'] = 'fold on: one kernel launch per planned call (+ a memset of 4 B per long row); off (the default): fused + combine'
            return fd

        if a.only_fold_leg:  # (the child process of the `fold` leg below: same graph, same plan, nothing else)
            print(json.dumps(fold_leg() if planned else dict(skipped='no plan for this shape')))
            return

        # parity spot-check on the exact tensors being timed: sampled rows (always including the longest) against an
        # fp64 torch gather-sum; every row is checked against the sequential reference in the cpu_baseline leg below,
        # and the full parity suite is tests/ -m gpu
        C, _ = step()
        if a.reduce in ('sum', 'mean'):
            deg = (rp[1:] - rp[:-1]).long()
            gsel = torch.Generator(device=dev)
            gsel.manual_seed(123)
            sel = torch.cat([torch.randint(0, Mloc, (2047,), generator=gsel, device=dev), deg.argmax().view(1)])

            def self_check(Cx):
                """max over sampled rows (always the longest) of the rel. error against an fp64 gather-sum, in units of the
                row's bar: 1e-5 - except for rows the schedule chains sequentially (hub rows, or every row in a strict run), which
                carry the REFERENCE's own sqrt(len) drift from the exact sum (1.2e-5 at 50 k nnz, profiles/
                r04_chain_error_by_length.txt): 4e-5 there; the all-rows `parity` block compares them with the sequential
                reference itself."""
                th = 0 if strict_alg else _capi.hub_threshold()
                worst_ = 0.0
                for r in sel.tolist()[-64:] + sel.tolist()[:256]:
                    s0, e0 = int(rp[r]), int(rp[r + 1])
                    if e0 == s0:
                        continue
                    ref = (val[s0:e0].double()[:, None] * X[col[s0:e0].long()].double()).sum(0)
                    if a.reduce == 'mean':
                        ref = ref / (e0 - s0)
                    chained = bool(strict_alg) or (th and e0 - s0 > th)
                    err = ((Cx[r].double() - ref).abs() / ref.abs().clamp_min(1e-6)).max().item()
                    worst_ = max(worst_, err / (4.0 if chained else 1.0))
                return worst_

            worst = self_check(C)
            if worst >= 1e-5 and _capi.hub_threshold() and not strict_alg:
                # the hub chains (round 4) were verified on the CPU emulation only when this was written: if they misbehave on
                # this box, say so in the line and time the schedule without them rather than lose the measurement
                extra['hub_chain_fallback'] = f'self-check with hub chains on: scaled rel err {worst:.3e}; DGS_HUB_CHAIN=0 for this run'
                os.environ['DGS_HUB_CHAIN'] = '0'
                _capi.reload_tuning()
                step, planned = make_step(rp, col, val, X)
                C, _ = step()
                worst = self_check(C)
            if not worst < 1e-5:  # wrong results are REPORTED with the number, loudly - a line beats a traceback for whoever reads the run
                extra['parity_failed'] = f'bench self-check FAILED: scaled rel err {worst} (in units of the bar: 1e-5, 4e-5 on chained rows); the value below is of a WRONG result'
                print('bench.py: ' + extra['parity_failed'], file=sys.stderr)
            extra['self_check_max_rel_err_vs_fp64'] = worst
            extra['self_check_bar'] = ('value above is in units of the bar per row: rel. error vs an fp64 gather-sum / 1.0 for rows the '
                                       'schedule folds with a tree (bar 1e-5), / 4.0 for rows it CHAINS sequentially (hub rows, strict '
                                       'runs: bar 4e-5 vs fp64, because the reference\'s own chain is ~1.2e-5 from fp64 at 50 k nnz); '
                                       'pass = < 1e-5.  The contract check is `parity` (every element vs the sequential reference)')
        if a.reduce == 'sum' and not a.no_cpu_baseline:
            C_check = C.cpu().numpy()
        del C
        b_alg = alg_bytes_spmm(Mloc, K, N, nnz_total, True, a.reduce in ('max', 'min'))
        parallelism = 'single'
        workload = f'synthetic power-law CSR {Mloc}x{K}, nnz={nnz_total} (~{nnz_total / Mloc:.1f}/row, alpha={a.alpha}, ' \
                   f'max_deg={st["max_deg"]}), cols={a.cols}, SpMM-{a.reduce} feat={N}, fp32 values'
        extra['schedule'] = _capi.spmm_schedule(op, Mloc, K, N, nnz_total) + ('+plan' if planned else '') + \
            (f'+strict-{a.strict}' if strict_alg else '')
        if a.reduce in ('sum', 'mean') and not strict_alg:
            th = _capi.hub_threshold()
            dg = (rp[1:] - rp[:-1]).long()
            if th and int((dg > th).sum()) > 0:
                extra['schedule'] += '+hub'
            extra['hub_chain'] = dict(threshold=th, rows=int((dg > th).sum()) if th else 0,
                                      nnz=int(dg[dg > th].sum()) if th else 0,
                                      note='rows above the threshold are one sequential fmaf chain per feature (like rows <= 64 '
                                           'nnz); rows in between a fixed tree; DGS_HUB_CHAIN')
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
        extra['halo'] = dict(rows_per_gpu=int(eng.n_halo), bytes_per_gpu=int(eng.n_halo) * N * 4, locality=a.locality,
                             cols=a.cols)
        extra['imbalance'] = eng.imbalance()  # per-rank nnz / rows / halo in / rows out: max, mean, max over mean
        # the exchange alone (pack + all-to-all-v, nothing overlapped): what the step would cost if compute were free
        ex_steps = max(5, a.steps // 5)
        wall_x, _ = time_steps(lambda: eng.exchange(Xloc), ex_steps, 2, pg)
        tx = torch.tensor([wall_x], device=dev, dtype=torch.float64)
        if pg:
            allreduce_max(tx)
        extra['exchange_only_ms'] = round(tx.item() / ex_steps * 1e3, 4)
        workload = f'synthetic power-law CSR {K}x{K} ({Mloc} rows/GPU, ~{a.deg}/row), cols={a.cols}, ' \
                   f'locality={a.locality}, SpMM-{a.reduce} feat={N}, 1-D row partition + halo all-to-all-v'

    for _ in range(max(0, a.settle)):  # clocks / power state: see --settle; outside the W + K protocol below
        step()
    torch.cuda.synchronize()
    wall, ev = time_steps(step, a.steps, a.warmup, pg)
    t = torch.tensor([wall, ev], device=dev, dtype=torch.float64)
    if pg:
        # per-rank step times (min / max over ranks): a straggler or an unbalanced partition shows here, not in the max alone
        mine = torch.tensor([wall / a.steps * 1e3], dtype=torch.float64, device='cpu' if one_gpu else dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        per_rank = [float(x.item()) for x in allr]
        extra['per_rank_ms'] = dict(min=round(min(per_rank), 5), max=round(max(per_rank), 5),
                                    ranks=[round(x, 5) for x in per_rank])
        allreduce_max(t)
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
        'settle_steps': max(0, a.settle),
        'ms_per_step': round(ms, 5),
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': workload, 'rows_per_gpu': Mloc, 'nnz_total': int(nnz_total), 'feat': N,
                   'reduce': a.reduce, 'cols': a.cols, 'seed': a.seed, 'parallelism': parallelism},
        'roofline': {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': None, 'traffic_source': None,
                     'alg_bytes_per_launch': int(b_alg), 'kernel_us': round(kern_s * 1e6, 2)},
    }
    res.update(extra)
    res['device_gate'] = dict(hub_chains=_capi.hub_gate(), in_kernel_fold=_capi.fold_gate(), hub_threshold=_capi.hub_threshold(),
                              note='hub_chains: 1 = dgs_spmm_hub_selftest passed on this device (chains on by default), -1 = failed (off), '
                                   '0 = not run; DGS_HUB_CHAIN overrides.  in_kernel_fold: verdict of dgs_spmm_fold_selftest (run by the '
                                   '`fold` leg below, after this snapshot unless DGS_FOLD=2); the fold itself is OFF unless DGS_FOLD=1 | 2')
    # roofline.traffic is NOT measured by this process (PMC passes cannot run beside the timed region): it is the
    # figure of the committed rocprofv3 counter passes for this very configuration, with its provenance next to it
    tf = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    if os.path.exists(tf) and not use_dist:
        try:
            tj = json.load(open(tf))
            sched_s = res.get('schedule', '')
            key = f'{a.reduce}_feat{N}_{a.cols}' + ('_plan' if '+plan' in sched_s else '') + \
                ('_hub' if '+hub' in sched_s else '') + (f'_strict_{a.strict}' if a.strict else '')
            if key in tj:
                res['roofline']['traffic'] = tj[key]
                res['roofline']['traffic_source'] = f"profiles/hbm_traffic.json[{key}] <- {tj.get('_source', {}).get(key, 'see profiles/README.md')}"
                # the counters were collected in another process: say so when this run's call time is not the one they belong to
                us0 = tj.get('_call_us', {}).get(key)
                if us0:
                    res['roofline']['traffic_call_us_at_collection'] = us0
                    res['roofline']['traffic_stale'] = bool(abs(kern_s * 1e6 - us0) > 0.07 * us0)
            else:
                res['roofline']['traffic_source'] = f'no counter pass committed for this schedule yet (profiles/hbm_traffic.json has no key {key})' 
        except Exception:
            pass

    try:  # a side leg must never lose the line (first hardware contact of a schedule may be this very run)
        if not a.no_protocol and not use_dist:
            # median of 5 repeats of K launches between HIP events; unit weights; seeds 1..4
            reps = sorted(event_ms(step, a.steps) for _ in range(5))
            prot = {'repeats_ms': [round(x, 5) for x in reps], 'median_ms': round(reps[2], 5),
                    'median_frac': round(b_alg / (reps[2] / 1e3) / 1e9 / HBM_PEAK_GBS, 4)}
            ones = torch.ones_like(val)
            step1, _ = make_step(rp, col, ones, X)
            step1()
            prot['unit_weights_ms'] = round(sorted(event_ms(step1, max(10, a.steps // 5)) for _ in range(3))[1], 5)
            del ones, step1
            seeds = {}
            for s in range(max(1, a.protocol_seeds)):
                if s == a.seed:
                    seeds[str(s)] = dict(ms=prot['median_ms'], nnz=int(nnz_total))
                    continue
                rp2, col2, st2, val2, X2 = make_graph(s)
                st2p, _ = make_step(rp2, col2, val2, X2)
                C2, _ = st2p()
                seeds[str(s)] = dict(ms=round(sorted(event_ms(st2p, max(10, a.steps // 5)) for _ in range(3))[1], 5),
                                     nnz=int(st2['nnz']))
                if a.reduce == 'sum' and not a.no_cpu_baseline:
                    # the 1e-5 claim is a property of the schedule, not of seed 0 (VERDICT r4 #2a): every element of every seed
                    # against the reference's sequential chain (the oracle's mul-add chain, pinned bit for bit to
                    # spmm_reference_host by tests/test_oracle_pin.py, on all host cores)
                    seeds[str(s)]['parity'] = parity_vs_sequential(C2.cpu().numpy(), rp2.cpu().numpy(), col2.cpu().numpy(),
                                                                   val2.cpu().numpy(), X2.cpu().numpy())
                del rp2, col2, val2, X2, st2p, C2
            prot['seeds'] = seeds
            res['protocol'] = prot
            if planned:
                # the in-kernel fold leg (fold_leg above).  In a CHILD process by default: it forces an opt-in schedule that has
                # never met the hardware, and whatever happens to it - a wrong bit, a hang cut by the timeout, a GPU fault that
                # kills the process - the line of the default schedule survives
                if a.fold_inprocess:
                    res['fold'] = fold_leg()
                else:
                    import subprocess
                    argv = [x for x in sys.argv[1:] if x not in ('--sweep',)] + ['--only-fold-leg', '--no-cpu-baseline', '--no-dense', '--no-protocol']
                    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
                    try:
                        p = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv, capture_output=True, text=True, timeout=600, env=env)
                        lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
                        res['fold'] = json.loads(lines[-1]) if lines else dict(error='the fold leg printed no line', rc=p.returncode,
                                                                                stderr_tail=p.stderr[-400:])
                    except subprocess.TimeoutExpired:
                        res['fold'] = dict(error='the fold leg did not finish within 600 s (killed)')
                    res['fold']['process'] = 'child'
            if '+hub' in res.get('schedule', ''):
                # what the hub chains cost next to the tree on the same tensors (the decision rule needs both numbers
                # in every line): DGS_HUB_CHAIN=0 for one measurement, then back to what this process was started with
                had = os.environ.get('DGS_HUB_CHAIN')
                os.environ['DGS_HUB_CHAIN'] = '0'
                _capi.reload_tuning()
                step0, _ = make_step(rp, col, val, X)
                step0()
                off_ms = sorted(event_ms(step0, max(10, a.steps // 5)) for _ in range(3))[1]
                del step0
                if had is None:
                    os.environ.pop('DGS_HUB_CHAIN')
                else:
                    os.environ['DGS_HUB_CHAIN'] = had
                _capi.reload_tuning()
                res['hub_chain'].update(on_ms=prot['median_ms'], off_ms=round(off_ms, 5),
                                        cost_frac=round(prot['median_ms'] / off_ms - 1.0, 4),
                                        rule='chains stay the default while they cost <= 5 % over the tree (DESIGN.md 4.1g)')
    except Exception as e:  # noqa: BLE001
        res.setdefault('leg_errors', {})['protocol_fold_hub'] = repr(e)

    C_strict = {}
    try:  # a side leg must never lose the line (first hardware contact of a schedule may be this very run)
        if not a.no_protocol and not use_dist:
            # what the plan costs and what it buys: blocking C-ABI build incl. compaction, and the plan-free step
            if planned:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _capi.spmm_plan(rp, col, K, N, force=(a.plan == 1))
                torch.cuda.synchronize()
                build_ms = (time.perf_counter() - t0) * 1e3
                pfree = sorted(event_ms(lambda: _capi.spmm(op, rp, col, val, X), max(10, a.steps // 5)) for _ in range(3))[1]
                res['plan'] = dict(build_ms_blocking=round(build_ms, 3), planfree_ms_per_step=round(pfree, 5),
                                   planned_ms_per_step=res['protocol']['median_ms'],
                                   calls_to_break_even_blocking=round(build_ms / max(pfree - res['protocol']['median_ms'], 1e-6), 1),
                                   note='dgsparse.Storage queues the build on the caller\'s stream at the 4th use of a matrix '
                                        '(DGS_PLAN_AFTER), with provisional counts and no host synchronisation (DESIGN.md 4.1d)')
            # the strict-order schedule on the same tensors (opt-in `algorithm` bits): time + full parity in cpu_baseline
            if a.reduce in ('sum', 'mean'):
                sd = {}
                plan_s = _capi.spmm_plan(rp, col, K, N, force=(a.plan == 1)) if planned else None
                for mode, alg in (('fma', _capi.ALG_STRICT_SUM), ('nofma', _capi.ALG_STRICT_NOFMA)):
                    fn = (lambda alg=alg: _capi.spmm(op, rp, col, val, X, algorithm=alg, plan=plan_s))
                    Cs, _ = fn()
                    if a.reduce == 'sum' and not a.no_cpu_baseline:
                        C_strict[mode] = Cs.cpu().numpy()
                    del Cs
                    ms_s = sorted(event_ms(fn, max(10, a.steps // 5)) for _ in range(3))[1]
                    sd[mode] = dict(ms_per_step=round(ms_s, 5), gflops=round(flops / (ms_s / 1e3) / 1e9, 1),
                                    frac=round(b_alg / (ms_s / 1e3) / 1e9 / HBM_PEAK_GBS, 4))
                sd['algorithm_bits'] = dict(fma=hex(_capi.ALG_STRICT_SUM), nofma=hex(_capi.ALG_STRICT_NOFMA))
                sd['over_the_plan'] = plan_s is not None
                res['strict'] = sd
            # the PUBLIC operator on the same graph (reference harness times this: benchmark/bench_spmm_time.py:35-45)
            try:
                fnp = {'sum': dgsparse.spmm_sum, 'mean': dgsparse.spmm_mean, 'max': dgsparse.spmm_max, 'min': dgsparse.spmm_min}[a.reduce]
                A_pub = dgsparse.SparseTensor(rowptr=rp, col=col, values=val, has_value=True)
                A_pub.storage.spmm_plan('csr', N, wait=True)
                A_pub.storage.spmm_plan('csc', N, wait=True)
                with torch.no_grad():
                    fnp(A_pub, X, 0)
                    pub_ng = sorted(event_ms(lambda: fnp(A_pub, X, 0), max(10, a.steps // 5)) for _ in range(3))[1]
                Xg = X.clone().requires_grad_()

                def fb():
                    o = fnp(A_pub, Xg, 0)
                    o.backward(o)
                    Xg.grad = None
                fb()
                pub_fb = sorted(event_ms(fb, max(5, a.steps // 10)) for _ in range(3))[1]
                res['public_api_ms'] = dict(op=f'dgsparse.spmm_{a.reduce}', no_grad=round(pub_ng, 5), fwd_bwd_dense_grad=round(pub_fb, 5),
                                            c_abi_step=res['protocol']['median_ms'])
                del A_pub, Xg
            except Exception as e:  # noqa: BLE001
                res['public_api_ms'] = dict(error=repr(e))
    except Exception as e:  # noqa: BLE001
        res.setdefault('leg_errors', {})['plan_strict_public_api'] = repr(e)

    if pg and use_dist and not a.no_worst_case:
        # the same step on the exchange's worst case: uniform random columns, no locality (every edge leaves its
        # partition with probability (N-1)/N); extra keys, not the metric - and never allowed to lose the line above
        try:
            from dgsparse import dist as ddist
            del eng
            part_w = ddist.synthetic_partition(rank, world, Mloc, a.deg, cols='uniform', locality=1.0 / world,
                                               seed=a.seed, device=dev)
            eng_w = ddist.DistSpMM(part_w, N)
            Xw = eng_w.local_features()
            Xw.copy_(torch.rand((Mloc, N), device=dev))
            ws, ww = max(5, a.steps // 5), 3
            wall_w, _ = time_steps(lambda: eng_w.spmm(Xw, a.reduce), ws, ww, True)
            tw = torch.tensor([wall_w, float(eng_w.n_halo)], device=dev, dtype=torch.float64)
            allreduce_max(tw)
            res['worst_case'] = dict(cols='uniform', locality=round(1.0 / world, 4), steps=ws,
                                     ms_per_step=round(tw[0].item() / ws * 1e3, 4),
                                     value=round(2.0 * eng_w.global_nnz * N / (tw[0].item() / ws) / 1e9, 2),
                                     unit='GFLOP/s', halo_rows_per_gpu_max=int(tw[1].item()),
                                     halo_bytes_per_gpu_max=int(tw[1].item()) * N * 4)
            del eng_w, part_w
        except Exception as e:  # noqa: BLE001
            res['worst_case'] = dict(error=repr(e))

    try:  # a side leg must never lose the line (first hardware contact of a schedule may be this very run)
        if a.sweep and not use_dist and rank == 0:
            sw = {}
            for n2, red in ((32, 'sum'), (128, 'sum'), (64, 'max'), (64, 'mean')):
                X2 = torch.rand((K, n2), device=dev)
                o2 = {'sum': _capi.SUM, 'max': _capi.MAX, 'mean': _capi.MEAN}[red]
                plan2 = _capi.spmm_plan(rp, col, K, n2) if (a.plan != 0 and hasattr(_capi, 'spmm_plan')) else None
                kw = dict(plan=plan2) if plan2 is not None else {}
                w2, e2 = time_steps(lambda: _capi.spmm(o2, rp, col, val, X2, **kw), 50, 10, False)  # 20 + 3 read 4 % slow (fresh operand, short run)
                b2 = alg_bytes_spmm(Mloc, K, n2, nnz_total, True, red == 'max')
                sw[f'{red}_feat{n2}'] = dict(gflops=round(2.0 * nnz_total * n2 / (w2 / 50) / 1e9, 1),
                                             gbs=round(b2 / (e2 / 50) / 1e9, 1), frac=round(b2 / (e2 / 50) / 1e9 / HBM_PEAK_GBS, 4))
            res['sweep'] = sw
    except Exception as e:  # noqa: BLE001
        res.setdefault('leg_errors', {})['sweep'] = repr(e)

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
                alg_gbs=round(b3 / (e3 / 10) / 1e9, 1), frac=round(b3 / (e3 / 10) / 1e9 / HBM_PEAK_GBS, 4),
                alg_bytes_per_launch=int(b3),
                # this configuration's compulsory intensity (~25 flop/B, B fits the Infinity Cache) puts it
                # under the L2 gather rate, not under HBM: one 512-byte row of B per nnz through the CUs' vector-memory
                # pipelines, measured at 20.6 TB/s chip-wide for L2 hits (experiments/gather_policy.cpp)
                gather=dict(bytes_per_launch=int(st3['nnz']) * 128 * 4, achieved_tbs=round(st3['nnz'] * 512 / (e3 / 10) / 1e12, 2),
                            l2_gather_rate_tbs=20.6, frac=round(st3['nnz'] * 512 / (e3 / 10) / 1e12 / 20.6, 3)))
            try:
                tj3 = json.load(open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json')))
                if 'C3_sum_feat128_panel' in tj3:
                    res['dense_graph']['traffic'] = tj3['C3_sum_feat128_panel']
                    res['dense_graph']['traffic_source'] = 'profiles/hbm_traffic.json[C3_sum_feat128_panel] <- ' + \
                        tj3.get('_source', {}).get('C3_sum_feat128_panel', 'see profiles/README.md')
            except Exception:
                pass
            del rp3, col3, val3, X3
        except Exception as e:
            res['dense_graph'] = dict(error=str(e))

    if rank == 0 and not use_dist and not a.no_cpu_baseline:
        try:
            res.update(cpu_baseline(rp.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy(), X.cpu().numpy(), flops,
                                    C_check, C_strict))
        except Exception as e:  # the baseline is a reported side figure; never lose the GPU line over it
            res['cpu_baseline'] = dict(value=None, unit='GFLOP/s', cores=0, kind='port', sample=f'failed: {e}')
    # A result that failed the self-check or the contract check must not be scoreable: the timing moves to
    # `unscored`, `value` and the roofline figures become null, and the process exits non-zero after the line is out.
    wrong = res.get('parity_failed') is not None
    if wrong:
        res['unscored'] = dict(value=res['value'], ms_per_step=res['ms_per_step'], roofline_achieved=res['roofline']['achieved'],
                               roofline_frac=res['roofline']['frac'], note='timing of a result that FAILED its parity check')
        res['value'] = None
        res['roofline']['achieved'] = None
        res['roofline']['frac'] = None
    if rank == 0:
        print(json.dumps(res))
    if pg:
        torch.distributed.destroy_process_group()
    if wrong:
        sys.exit(3)


if __name__ == '__main__':
    main()
