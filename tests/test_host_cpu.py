"""CPU-side tests: the C-ABI library loads and exports every symbol include/dgsparse_hip.h declares, the
Python surface mirrors the reference's names, host-side validation fires, and nothing in the product
package touches oracle/.  No compute calls (there is no GPU here)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'dgsparse-lib_amd')


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'dgsparse_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(dgs_\w+|gespmm\w+|csrspmm_\w+|spmm_cuda\w*|sddmm_cuda_\w+)\s*\(', hdr))
    assert len(declared) >= 16
    lib = ctypes.CDLL(os.path.join(PKG, 'dgsparse', 'libdgsparse_hip.so'))
    for sym in sorted(declared):
        assert hasattr(lib, sym), f'{sym} declared in include/dgsparse_hip.h but not exported'
    from dgsparse import _capi
    assert set(_capi.EXPORTS) == declared
    assert _capi.arch() == 'gfx950' and _capi.version() >= 1000
    lib.dgs_strerror.restype = ctypes.c_char_p
    assert lib.dgs_strerror(-2) == b'workspace too small'


def test_python_surface_matches_reference_names():
    import dgsparse
    for name in ['spmm_sum', 'spmm_max', 'spmm_min', 'spmm_mean', 'Storage', 'SparseTensor', 'csr2csc']:
        assert hasattr(dgsparse, name)  # reference dgsparse/__init__.py:46-49
    assert dgsparse._C.cuda_version() == -1
    for op in ['spmm_sum', 'spmm_max', 'spmm_min', 'spmm_mean', 'csr2csc', 'sddmm']:
        assert hasattr(torch.ops.dgsparse_spmm, op)
    schema = str(torch.ops.dgsparse_spmm.spmm_sum.default._schema)
    # same 9 positional arguments as src/spmm.cpp:83-89 (the C++ binding infers names _0.._8 like the reference's)
    assert schema.count('Tensor') == 8 and 'bool' in schema and 'int' in schema


def test_no_cpu_fallback_and_validation():
    import dgsparse
    from dgsparse import _capi
    rp = torch.tensor([0, 1, 2], dtype=torch.int)
    col = torch.tensor([0, 1], dtype=torch.int)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _capi.spmm(0, rp, col, None, torch.rand(2, 4))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        dgsparse.SparseTensor(rowptr=rp, col=col)
    with pytest.raises(AssertionError):
        dgsparse.Storage(rowptr=rp.long(), col=col)  # reference storage.py:54
    with pytest.raises(AssertionError):
        dgsparse.Storage(rowptr=rp, col=col.long())  # reference storage.py:29
    st = dgsparse.Storage.empty()  # reference storage.py:103
    assert st.sparse_sizes == (0, 0) and st.nnz == 0
    # pre-supplied CSC arrays skip the conversion (storage.py:160-161), so host logic is testable on CPU
    st = dgsparse.Storage(rowptr=rp, col=col, row=torch.tensor([0, 1], dtype=torch.int),
                          colptr=torch.tensor([0, 1, 2]), csr2csc=torch.tensor([0, 1]))
    assert st.colptr().dtype == torch.int and st.csr2csc().tolist() == [0, 1]
    assert st.values().tolist() == [1.0, 1.0]  # defaults to ones (storage.py:60-68)
    st._colptr = None
    with pytest.raises(ValueError):
        st.colptr()


def test_product_package_never_touches_the_oracle():
    bad = []
    for dp, _, fs in os.walk(PKG):
        for f in fs:
            if f.endswith(('.py', '.hip', '.h', '.cpp', 'Makefile')):
                txt = open(os.path.join(dp, f), errors='ignore').read()
                if re.search(r'^\s*(import|from)\s+oracle\b', txt, flags=re.M) or 'liborc' in txt or 'oracle/' in txt:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def _write_mtx(path, n, entries, symmetric, field='real'):
    with open(path, 'w') as f:
        f.write(f'%%MatrixMarket matrix coordinate {field} {"symmetric" if symmetric else "general"}\n% comment\n')
        f.write(f'{n} {n} {len(entries)}\n')
        for r, c in entries:
            f.write(f'{r + 1} {c + 1}' + ('' if field == 'pattern' else ' 1.5') + '\n')


@pytest.mark.parametrize('symmetric,field', [(False, 'real'), (True, 'pattern'), (True, 'real')])
def test_read_mtx_matches_reference_loader(tmp_path, symmetric, field):
    """dgsparse.io.read_mtx against the reference's own read_mtx_file (oracle/_ref) where that was built, and
    against a direct construction everywhere."""
    import numpy as np

    import oracle
    from dgsparse import io as dio
    rng = np.random.default_rng(3)
    n = 60
    ent = [(int(a), int(b)) for a, b in rng.integers(0, n, (300, 2))]
    if symmetric:
        ent = sorted({(max(a, b), min(a, b)) for a, b in ent})  # lower triangle, unique (as mtx files store it)
    p = str(tmp_path / 'g.mtx')
    _write_mtx(p, n, ent, symmetric, field)
    nrow, ncol, rp, col = dio.read_mtx(p)
    keys = [a * n + b for a, b in ent]
    if symmetric:
        keys = sorted(set(keys) | {b * n + a for a, b in ent})
    else:
        keys = sorted(keys)
    assert nrow == n and ncol == n and rp[-1] == len(keys)
    assert np.array_equal(col, np.array([k % n for k in keys], np.int32))
    assert np.array_equal(np.diff(rp), np.bincount([k // n for k in keys], minlength=n))
    if oracle.have_ref():
        r2, c2, rp2, col2 = oracle.ref_read_mtx(p)
        assert (r2, c2) == (nrow, ncol)
        if symmetric:
            # reference quirk (sp_util.hpp:238-247): the CSR loop is bounded by the FILE's entry count, not by the
            # mirrored list, so only the first `nnz_file` mirrored entries survive and later rows come out empty.
            # The build keeps the whole mirrored matrix; the reference output must be exactly its prefix.
            nf = len(ent)
            assert np.array_equal(col2, col[:nf]) and np.array_equal(rp2, np.minimum(rp, nf))
        else:
            assert np.array_equal(rp2, rp) and np.array_equal(col2, col)


def test_read_mtx_reference_fixture():
    import numpy as np

    import oracle
    path = '/root/reference/example/data/p2p-Gnutella31.mtx'
    if not (os.path.exists(path) and oracle.have_ref()):
        pytest.skip('reference tree not present')
    from dgsparse import io as dio
    nrow, ncol, rp, col = dio.read_mtx(path)
    r2, c2, rp2, col2 = oracle.ref_read_mtx(path)
    assert (nrow, ncol) == (r2, c2) and np.array_equal(rp, rp2) and np.array_equal(col, col2)


def test_schedule_queries_need_no_gpu(monkeypatch):
    """dgs_spmm_csr_schedule / dgs_sddmm_csr_schedule are pure functions of the sizes (and of the DGS_PANEL overrides):
    the headline graph stays on the row-stream schedule, the Reddit-shaped one takes the column-panel sweep, odd
    feature widths only after padding, small inputs take the single launch."""
    from dgsparse import _capi
    for k in ('DGS_PANEL', 'DGS_PANEL_KB', 'DGS_PANEL_LEAD', 'DGS_PANEL_TLONG'):
        monkeypatch.delenv(k, raising=False)
    M1, nnz1 = 1 << 20, 16_108_469            # bench.py headline: 16 nnz per row
    Mr, nnzr = 232_965, 114_609_723           # Reddit-shaped: 492 nnz per row
    assert _capi.spmm_schedule(_capi.SUM, 2708, 2708, 32, 10_556) == 'small'
    assert _capi.spmm_schedule(_capi.SUM, M1, M1, 64, nnz1) == 'rows'
    for op in (_capi.SUM, _capi.MEAN, _capi.MAX, _capi.MIN):
        for n in (32, 64, 128, 256, 512):  # max/min at 256+ hold few rows per sweep but still qualify (long visits)
            assert _capi.spmm_schedule(op, Mr, Mr, n, nnzr) == 'panel', (op, n)
    assert _capi.spmm_schedule(_capi.SUM, Mr, Mr, 16, nnzr) == 'rows'      # 4 lanes per row: not a panel shape
    assert _capi.spmm_schedule(_capi.SUM, Mr, Mr, 41, nnzr) == 'rows'      # not a multiple of 4 ...
    assert _capi.spmm_schedule(_capi.SUM, Mr, Mr, 44, nnzr) == 'panel'     # ... the bindings pad to this
    assert _capi.spmm_schedule(_capi.SUM, Mr, Mr, 128, Mr * 64) == 'rows'  # 64 nnz per row: too little reuse
    assert _capi.spmm_schedule(_capi.SUM, Mr, Mr, 128, Mr * 128) == 'rows'  # reuse 4.5 but only 6.7 nnz per visit
    assert _capi.spmm_schedule(_capi.SUM, Mr, Mr, 128, Mr * 160) == 'panel'  # reuse 5.6
    lib = _capi._lib
    assert lib.dgs_sddmm_csr_schedule(Mr, Mr, 64, nnzr, 0) == 2 and lib.dgs_sddmm_csr_schedule(Mr, Mr, 64, nnzr, 1) == 2
    assert lib.dgs_sddmm_csr_schedule(M1, M1, 64, nnz1, 0) == 1
    monkeypatch.setenv('DGS_PANEL', '0')
    assert _capi.spmm_schedule(_capi.SUM, Mr, Mr, 128, nnzr) == 'rows'
    monkeypatch.setenv('DGS_PANEL', '1')
    assert _capi.spmm_schedule(_capi.SUM, M1, M1, 64, nnz1) == 'panel'



def test_plan_host_helpers_need_no_gpu():
    """The host-only parts of the non-blocking plan build (include/dgsparse_hip.h): thresholds, provisional (upper-bound)
    counts from four sums over the row lengths, and the parser of a host copy of the plan header."""
    from dgsparse import _capi
    t1, ts = _capi.plan_thresholds()
    assert t1 == 64 and ts >= 128
    info = _capi.plan_provisional_info(1 << 24, 25000, 9_000_000, 5500, 6_700_000)
    n_units, n_long, n_pslots = int(info[0]), int(info[1]), int(info[2])
    assert n_long == 25000 and n_units == n_pslots and n_units >= 5500 * 9  # every cut row may have 8 cells + 1
    assert int(info[14]) == 0, 'off_long = 0: the build-time layout'
    # more long rows / nnz never shrink the bound
    bigger = _capi.plan_provisional_info(1 << 24, 26000, 9_500_000, 5600, 6_900_000)
    assert int(bigger[0]) >= n_units and int(bigger[1]) >= n_long
    for bad in ((0, 1, 1, 0, 0), (100, 1, 10, 2, 5), (100, 2, 5, 1, 10)):  # nnz <= 0; tslice rows > t1 rows; tslice nnz > t1 nnz
        with pytest.raises(RuntimeError):
            _capi.plan_provisional_info(*bad)
    pi = _capi.PlanInfo()
    junk = (ctypes.c_char * 256)()
    assert _capi._lib.dgs_spmm_plan_info_from_header(junk, 256, ctypes.byref(pi)) == -1, 'no magic: DGS_EINVAL'
    assert _capi._lib.dgs_spmm_plan_info_from_header(junk, 16, ctypes.byref(pi)) == -1, 'short buffer'
    hdr = (ctypes.c_int32 * 64)()
    hdr[0], hdr[1] = 0x64677350, 2          # magic "dgsP", version
    hdr[5], hdr[6], hdr[7] = 1234, 56, 789  # n_units, n_long, n_pslots
    hdr[10] = 256                           # tslice
    for x in range(9):
        hdr[12 + x] = 100 * x               # xcd_start
    assert _capi._lib.dgs_spmm_plan_info_from_header(hdr, 256, ctypes.byref(pi)) == 0
    assert (pi.n_units, pi.n_long, pi.n_pslots, pi.tslice, pi.off_long) == (1234, 56, 789, 256, 0)
    assert list(pi.xcd_start) == [100 * x for x in range(9)]


def test_nnz_balanced_row_offsets():
    """dgsparse.dist.row_offsets: contiguous blocks with equal nnz (block g ends at the first row whose prefix nnz reaches
    (g+1)*nnz/P), monotone with runs of empty rows, degenerate sizes."""
    import numpy as np
    from dgsparse import dist as dd
    rng = np.random.default_rng(0)
    deg = np.minimum(rng.zipf(1.5, 5000), 3000)
    deg[rng.integers(0, 5000, 800)] = 0
    deg = np.sort(deg)[::-1].copy()  # degree-correlated order: equal row blocks give rank 0 the hubs
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    for world in (1, 2, 3, 8):
        offs = dd.row_offsets(torch.from_numpy(rp), world, 'nnz')
        assert offs[0] == 0 and offs[-1] == 5000 and all(a <= b for a, b in zip(offs, offs[1:])) and len(offs) == world + 1
        shares = [int(rp[offs[i + 1]] - rp[offs[i]]) for i in range(world)]
        assert max(shares) - min(shares) <= 2 * int(deg.max()) + 1, shares
        rows_eq = dd.row_offsets(torch.from_numpy(rp), world, 'rows')
        eq_shares = [int(rp[rows_eq[i + 1]] - rp[rows_eq[i]]) for i in range(world)]
        if world > 1:
            assert max(shares) <= max(eq_shares)
    assert dd.row_offsets(torch.tensor([0, 0, 0]), 2, 'nnz') == [0, 0, 2]
    with pytest.raises(ValueError):
        dd.row_offsets(torch.from_numpy(rp), 2, 'cols')


def test_hub_work_deal_covers_every_task_once(tmp_path):
    """csrc/spmm_strict.h strict_deal + the hub-table regions, checked on the host (tests/host/deal_test.cpp): every (row, slice)
    task of every segment goes to exactly one worker on the row's XCD, for grids that are and are not multiples of 8."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    exe = str(tmp_path / 'deal_test')
    src = os.path.join(ROOT, 'tests', 'host', 'deal_test.cpp')
    r = subprocess.run([hipcc, '--offload-arch=gfx950', '-O1', '-std=c++17', '-I' + os.path.join(ROOT, 'dgsparse-lib_amd', 'csrc'),
                        '-I' + os.path.join(ROOT, 'include'), '-Wno-unused-function', src, '-o', exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == 'ok', r.stdout + r.stderr


def test_counter_based_sampler_matches_its_definition():
    """bench/graphgen.py hash_bits is splitmix64 of (seed, stream, index) written with integer tensor ops, so that the GPU and the
    CPU build the SAME graphs / values / features (bench.py, bench/emu_parity.py, tests/test_gpu_fullsize.py rely on it): check
    the words against plain Python integers, the unit floats against their definition, and a small graph against itself."""
    from bench import graphgen

    def sm(i, seed, stream):
        m = (1 << 64) - 1
        z = (i + ((seed * 0xD1B54A32D192ED03 + stream * 0x8CB92BA72F3D8DD7 + 0x9E3779B97F4A7C15) & m)) & m
        z = (z * 0x9E3779B97F4A7C15) & m
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
        z ^= z >> 31
        return z - (1 << 64) if z >= (1 << 63) else z

    for seed, stream, start in ((0, 1, 0), (3, 102, 12345), (4, 2, (1 << 33) + 7)):
        got = graphgen.hash_bits(6, seed, stream, start=start).tolist()
        assert got == [sm(start + i, seed, stream) for i in range(6)]
    u = graphgen.hash_unit(4, 1, 5).tolist()
    assert u == [((sm(i, 1, 5) % (1 << 64)) >> 11) / float(1 << 53) for i in range(4)]
    v = graphgen.values_t(4, 2).tolist()
    assert v == [((sm(i, 2, 101) % (1 << 64)) >> 40) / float(1 << 24) for i in range(4)]
    a = graphgen.powerlaw_csr(5000, 60000, alpha=2.0, dmax=700, seed=5, sampler='hash')
    b = graphgen.powerlaw_csr(5000, 60000, alpha=2.0, dmax=700, seed=5, sampler='hash')
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and a[2]['nnz'] > 55000
    for cols in ('uniform', 'local'):
        rp, col, st = graphgen.powerlaw_csr(3000, 30000, alpha=2.2, dmax=300, cols=cols, seed=1, sampler='hash')
        assert 0 <= col.min() and col.max() < 3000 and rp[-1] == col.shape[0]


def test_gate_verdict_cache_key_read_write(tmp_path, monkeypatch):
    """DGS_GATE_CACHE (the device gate is per process - every DataLoader worker and rank pays it): a PASS is kept
    per (library binary, device model, runtime) key; a different device model or library is a different key; failures are never
    written; a damaged or foreign file is no verdict."""
    from dgsparse import _capi
    props = dict(name='AMD Instinct MI355X', arch='gfx950:sramecc+:xnack-', cus=256, mem=309220868096, hip='7.0.0', abi=_capi.version())
    k1 = _capi._gate_key(props)
    assert len(k1) == 32 and k1 == _capi._gate_key(dict(props))
    assert _capi._gate_key(dict(props, cus=304)) != k1 and _capi._gate_key(dict(props, hip='7.2.0')) != k1
    d = str(tmp_path / 'gate')
    assert _capi._gate_cache_read(d, k1) is None            # no directory yet
    _capi._gate_cache_write(d, k1, props)
    assert _capi._gate_cache_read(d, k1) == 1
    assert _capi._gate_cache_read(d, _capi._gate_key(dict(props, cus=304))) is None
    with open(f'{d}/dgs_gate_{k1}.json', 'w') as f:
        f.write('{"key": "someone else", "hub_chains": 1}')
    assert _capi._gate_cache_read(d, k1) is None            # a file that does not carry its own key
    with open(f'{d}/dgs_gate_{k1}.json', 'w') as f:
        f.write('not json')
    assert _capi._gate_cache_read(d, k1) is None
    _capi._gate_cache_write('/proc/definitely/not/writable', k1, props)  # must not raise
