"""bench.py end to end WITHOUT a GPU: the whole flow of the line the driver parses - graph, plan, self-check, W + K protocol,
median-of-5, unit weights, seeds with their parity blocks, fold on / off, hub chains on / off, plan cost, strict modes, the
cpu_baseline / parity legs, the JSON schema - runs over the CPU emulation of the kernels (tests/emu) with the CUDA calls of
bench.py patched to host stand-ins (TEST INFRASTRUCTURE: nothing here is reachable from bench.py itself; the timings of such a
run mean nothing).  It exists because a typo in bench.py is only ever discovered on the GPU box otherwise - at round end."""
import io
import json
import os
import sys
import time
import types
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))
import emu_lib as E  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists('/opt/rocm/lib/llvm/bin/clang++'), reason='the emulation builds with ROCm\'s clang')


class _Plan:
    def __init__(self, pl):
        self.pl = pl
        self.info = pl[1]


def _np(t):
    return None if t is None else t.detach().cpu().contiguous().numpy()


def _shim():
    """The subset of dgsparse._capi that bench.py uses, over the emulated library (torch CPU tensors in and out)."""
    real = sys.modules['dgsparse._capi']
    m = types.SimpleNamespace(**{k: getattr(real, k) for k in ('SUM', 'MAX', 'MIN', 'MEAN', 'ALG_STRICT_SUM', 'ALG_STRICT_NOFMA',
                                                              'ALG_SHARED_GPU', 'ALG_NO_HUB_ROWS', 'ALG_NO_HUB_COLS')})
    m.calls = []

    def spmm(op, rp, col, val, X, algorithm=0, plan=None, **kw):
        m.calls.append((int(op), int(algorithm), plan is not None))
        rpn, coln, valn, Xn = _np(rp), _np(col), _np(val), _np(X)
        strict = int(algorithm) & (real.ALG_STRICT_SUM | real.ALG_STRICT_NOFMA)
        if strict:
            C = E.spmm_ex(int(op), rpn, coln, valn, Xn, algorithm=int(algorithm), plan=None if plan is None else plan.pl)
            return torch.from_numpy(C), None
        C, Eo = E.spmm(int(op), rpn, coln, valn, Xn, algorithm=int(algorithm), plan=None if plan is None else plan.pl)
        return torch.from_numpy(C), (None if Eo is None else torch.from_numpy(Eo))

    def spmm_plan(rp, col, K, N=64, force=False):
        rpn, coln = _np(rp), _np(col)
        if os.environ.get('DGS_PLAN', '1') == '0' and not force:
            return None
        if E.schedule(E.SUM, rpn.size - 1, int(K), int(N), coln.size) != 'rows':
            return None
        return _Plan(E.spmm_plan(rpn, coln, int(K)))

    m.spmm, m.spmm_plan = spmm, spmm_plan
    m.spmm_schedule = lambda op, M, K, N, nnz: E.schedule(int(op), int(M), int(K), int(N), int(nnz))
    m.hub_threshold = lambda: int(E.lib().dgs_spmm_hub_threshold())
    m.hub_gate = lambda: int(E.lib().dgs_spmm_hub_gate())
    m.fold_gate = lambda: int(E.lib().dgs_spmm_fold_gate())
    # (the full fold self-test emulates in ~75 s: the dry run walks the code with the line-sharing family only - a partial run,
    # so the gate stays where it was)
    m.fold_selftest = lambda **k: E.fold_selftest(rounds=1, load=False, families=[2])
    m.reload_tuning = lambda: E.lib().dgs_reload_tuning()
    return m


class _Event:
    def __init__(self, enable_timing=False):
        self.t = 0.0

    def record(self, *a):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3 + 1e-3

    def synchronize(self):
        pass

    def query(self):
        return True


@pytest.mark.parametrize('argv', [['--rows-log2', '15', '--dmax', '20000', '--alpha', '1.6'],
                                  ['--rows-log2', '15', '--dmax', '20000', '--alpha', '1.6', '--strict', 'fma', '--no-protocol']],
                         ids=['default line', '--strict fma'])
def test_bench_line_over_the_emulation(monkeypatch, argv):
    import dgsparse  # noqa: F401  (the real package loads without a GPU; only its _capi entry points are swapped below)
    import bench as bench_mod  # the bench/ package (graphgen) ...
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_py', os.path.join(ROOT, 'bench.py'))
    B = importlib.util.module_from_spec(spec)  # ... and bench.py itself
    spec.loader.exec_module(B)
    assert bench_mod is not None
    shim = _shim()
    real_capi = sys.modules['dgsparse._capi']
    for k in ('spmm', 'spmm_plan', 'spmm_schedule', 'hub_threshold', 'hub_gate', 'fold_gate', 'fold_selftest', 'reload_tuning'):
        monkeypatch.setattr(real_capi, k, getattr(shim, k))
    # CUDA stand-ins: every tensor bench.py makes lives on the CPU, events are wall-clock stamps
    real_device = torch.device
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 1)
    monkeypatch.setattr(torch.cuda, 'set_device', lambda *a: None)
    monkeypatch.setattr(torch.cuda, 'synchronize', lambda *a: None)
    monkeypatch.setattr(torch.cuda, 'Event', _Event)
    monkeypatch.setattr(torch, 'device', lambda *a, **k: real_device('cpu'))
    for k in list(os.environ):
        if k.startswith('DGS_'):
            monkeypatch.delenv(k)
    # (a 2^15-row graph of this generator has no row above the default hub threshold: lower it so that the line's hub-chain
    # legs - schedule tag, hub_chain.on_ms / off_ms - are walked as they are on the headline graph)
    monkeypatch.setenv('DGS_HUB_CHAIN', '1024')
    E.lib().dgs_reload_tuning()
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '1', '--steps', '2', '--warmup', '1', '--settle', '0', '--no-dense', '--protocol-seeds', '2', '--fold-inprocess'] + argv)
    # the protocol's repeat counts are for a GPU: one emulated launch per measurement is enough to walk the code
    monkeypatch.setattr(B, 'event_ms', lambda fn, steps: (fn(), 1.0)[1])
    out = io.StringIO()
    with redirect_stdout(out):
        B.main()
    line = [ln for ln in out.getvalue().splitlines() if ln.startswith('{')][-1]
    res = json.loads(line)
    # the contract's keys
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in res, k
    assert res['n_gpus'] == 1 and res['steps'] == 2 and res['warmup'] == 1 and res['dtype'] == 'f32' and res['vs_baseline'] is None
    assert set(res['roofline']) >= {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'}
    assert set(res['cpu_baseline']) >= {'value', 'unit', 'cores', 'kind', 'sample'} and res['cpu_baseline']['value']
    assert 'workload' in res['config']
    assert 'leg_errors' not in res and 'parity_failed' not in res, (res.get('leg_errors'), res.get('parity_failed'))
    assert res['device_gate']['hub_chains'] == 1 and res['device_gate']['in_kernel_fold'] in (0, 1)  # (the fold is opt-in: its gate only matters to DGS_FOLD=2)
    if '--strict' in argv:
        assert res['schedule'].endswith('+strict-fma') and 'rows+plan' in res['schedule']
        assert res['parity_strict']['fma']['bit_exact_vs_its_sequential_chain'] if 'parity_strict' in res else True
        return
    assert res['schedule'] == 'rows+plan+hub', res['schedule']
    assert res['hub_chain']['rows'] > 0 and res['hub_chain']['threshold'] == 1024
    assert 'on_ms' in res['hub_chain'] and 'off_ms' in res['hub_chain'] and os.environ['DGS_HUB_CHAIN'] == '1024'
    assert res['parity']['within_1e_5'] and res['parity']['elements_beyond_1e_5'] == 0
    assert {'on_ms', 'off_ms', 'gate', 'selftest', 'same_bits', 'default_on'} <= set(res['fold'])
    assert res['fold']['selftest'] == 1 and res['fold']['same_bits'] is True and res['fold']['default_on'] is False
    for s, v in res['protocol']['seeds'].items():
        if s != '0':
            assert v['parity']['within_1e_5'], (s, v)
    assert res['strict']['over_the_plan'] is True
    assert res['parity_strict']['fma']['bit_exact_vs_its_sequential_chain'] and res['parity_strict']['nofma']['bit_exact_vs_its_sequential_chain']
    assert 'self_check_bar' in res and res['self_check_max_rel_err_vs_fp64'] < 1e-5
