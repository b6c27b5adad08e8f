"""TEST INFRASTRUCTURE ONLY: builds (once per source change) and loads tests/emu/_build/libdgs_emu.so - the kernels of
dgsparse-lib_amd/csrc compiled as host code over the wave64 emulation of tests/emu/hip/hip_runtime.h - and wraps the C ABI
for numpy arrays (host memory IS device memory here)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SUM, MAX, MIN, MEAN = 0, 1, 2, 3
ALG_STRICT_SUM, ALG_STRICT_NOFMA, ALG_NO_HUB_ROWS = 0x200, 0x400, 0x800
_lib = None


class PlanInfo(ctypes.Structure):
    _fields_ = [('n_units', ctypes.c_int32), ('n_long', ctypes.c_int32), ('n_pslots', ctypes.c_int32), ('n_hub', ctypes.c_int32),
                ('tslice', ctypes.c_int32), ('xcd_start', ctypes.c_int32 * 9), ('off_long', ctypes.c_int32),
                ('off_hub', ctypes.c_int32)]


def lib():
    global _lib
    if _lib is None:
        asan = os.environ.get('DGS_EMU_ASAN') == '1'  # tests/emu/run_asan.sh: the sanitizer's runtime must be preloaded
        alt = os.environ.get('DGS_EMU_LIB')  # tests/emu/mutation_check.py: a library built from deliberately broken sources
        if alt:
            _lib = ctypes.CDLL(alt)
            for f in ('dgs_spmm_csr_workspace_bytes', 'dgs_spmm_plan_bytes', 'dgs_spmm_plan_workspace_bytes',
                      'dgs_spmm_csr_plan_workspace_bytes', 'dgs_spmm_plan_compact_bytes', 'dgs_spmm_hub_selftest_bytes'):
                getattr(_lib, f).restype = ctypes.c_size_t
            hub_selftest()  # (a mutant that breaks the hub workgroup fails it: the chains are then OFF unless DGS_HUB_CHAIN forces them)
            return _lib
        ubsan = os.environ.get('DGS_EMU_UBSAN') == '1'
        r = subprocess.run(['make', '-C', HERE, '-j8'] + (['ASAN=1'] if asan else ['UBSAN=1'] if ubsan else []), capture_output=True,
                           text=True)
        if r.returncode != 0:
            raise RuntimeError('emu build failed:\n' + r.stdout[-3000:] + r.stderr[-3000:])
        _lib = ctypes.CDLL(os.path.join(HERE, '_build_asan' if asan else '_build_ubsan' if ubsan else '_build', 'libdgs_emu.so'))
        for f in ('dgs_spmm_csr_workspace_bytes', 'dgs_spmm_plan_bytes', 'dgs_spmm_plan_workspace_bytes',
                  'dgs_spmm_csr_plan_workspace_bytes', 'dgs_spmm_plan_compact_bytes', 'dgs_spmm_hub_selftest_bytes'):
            getattr(_lib, f).restype = ctypes.c_size_t
        if os.environ.get('DGS_EMU_NO_SELFTEST') != '1':  # (tests of the gate itself load the library as a C caller would: unverified)
            assert hub_selftest() == 1, 'the hub-chain self-test fails on the emulation'
    return _lib


def launch_log(clear=True):
    """[(kernel expression as written at the launch site, grid.x, grid.y)] of every launch since the log was last cleared."""
    L = lib()
    L.emu_launch_log.restype = ctypes.c_size_t
    buf = ctypes.create_string_buffer(1 << 20)
    L.emu_launch_log(buf, ctypes.c_size_t(len(buf)), int(clear))
    out = []
    for line in buf.value.decode().splitlines():
        name, gx, gy = line.rsplit(' ', 2)
        out.append((name, int(gx), int(gy)))
    return out


def hub_selftest():
    """The device gate of the default hub chains (include/dgsparse_hip.h): what dgsparse's Python layer runs at the first use of
    a GPU, run here when the emulated library is loaded.  1 = passed (hub chains on by default), 0 = failed (off)."""
    nb = _lib.dgs_spmm_hub_selftest_bytes()
    scratch = _buf(nb)
    # (a process that pins DGS_HUB_CHAIN skips the test - include/dgsparse_hip.h; the emulation wants the verdict whatever a test
    # has set at load time)
    had = {k: os.environ.pop(k, None) for k in ('DGS_HUB_CHAIN', 'DGS_FOLD')}
    _lib.dgs_reload_tuning()
    try:
        return _lib.dgs_spmm_hub_selftest(_p(scratch), ctypes.c_size_t(nb), None)
    finally:
        for k, v in had.items():
            if v is not None:
                os.environ[k] = v
        _lib.dgs_reload_tuning()


def fold_selftest(rounds=1, load=False, families=None):
    """dgs_spmm_fold_selftest on the emulation: (verdict, mismatch count per family).  families: indices to run (None = all; only a
    full run moves the fold gate)."""
    L = lib()
    nb = L.dgs_spmm_hub_selftest_bytes()
    scratch = _ws(nb)  # (under DGS_EMU_MEM=relaxed the scratch - the products' workspaces live inside it - is the watched range)
    flags = (1 if load else 0) | (sum(1 << (8 + f) for f in families) if families is not None else 0)
    rc = L.dgs_spmm_fold_selftest(_p(scratch), ctypes.c_size_t(nb), int(rounds), int(flags), None)
    return rc, selftest_detail()[2:2 + L.dgs_spmm_selftest_families()]


def selftest_detail():
    """The mismatch counters of the last self-tests: [0] hub, [1] fold, [2 + f] fold family f, [16 + h] hub shape h."""
    out = (ctypes.c_int32 * 64)()
    lib().dgs_spmm_selftest_detail(out, 64)
    return list(out)


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def _buf(nbytes):
    return np.zeros(max(int(nbytes), 16) + 64, np.uint8)  # numpy's allocations are 64-byte aligned


def _ws(nbytes):
    """a call's workspace: under DGS_EMU_MEM=relaxed it is THE watched range of the emulator's memory model (partial rows, unit
    tables, arrival counters: where workgroups hand data to one another)"""
    ws = _buf(nbytes)
    if os.environ.get('DGS_EMU_MEM') == 'relaxed':
        mem_watch(None), mem_watch(ws)
    return ws


def set_env(**kw):
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    lib().dgs_reload_tuning()


last_ws = None  # the workspace of the last spmm() call (tests look at what a launch left in it)


def mem_watch(buf=None):
    """Relaxed-memory mode (DGS_EMU_MEM=relaxed in the environment BEFORE the library's first launch): watch this numpy buffer -
    the accesses the kernels make to it through load_vec / store_vec, the raw-buffer builtins and the agent-scope atomics go through
    the emulator's memory model (tests/emu/emu_rt.cpp).  None: forget every range."""
    if buf is None:
        lib().emu_mem_watch(None, ctypes.c_size_t(0))
    else:
        lib().emu_mem_watch(_p(buf), ctypes.c_size_t(buf.nbytes))


def mem_report(clear=True):
    """dict(unperformed=reads that missed a store still queued in another workgroup / dirty in another XCD's L2, l1_stale=plain
    loads served from a line pinned before a newer write-through, loads, stores) since the counters were last cleared."""
    out = (ctypes.c_ulonglong * 4)()
    lib().emu_mem_report(out, int(clear))
    return dict(unperformed=int(out[0]), l1_stale=int(out[1]), loads=int(out[2]), stores=int(out[3]))


def spmm(op, rp, col, val, X, algorithm=0, plan=None):
    L = lib()
    M, nnz, (K, N) = rp.size - 1, col.size, X.shape
    C = np.full((M, N), np.nan, np.float32)
    E = np.full((M, N), -7, np.int32) if op in (MAX, MIN) else None
    i64 = ctypes.c_int64
    if plan is not None:
        pbuf, info = plan
        wsb = L.dgs_spmm_csr_plan_workspace_bytes(op, i64(M), i64(N), i64(nnz), ctypes.byref(info))
        ws = _ws(wsb)
        rc = L.dgs_spmm_csr_plan_f32(op, i64(M), i64(K), i64(N), i64(nnz), _p(rp), _p(col), _p(val), _p(X), _p(C), _p(E), _p(pbuf),
                                     ctypes.byref(info), _p(ws), ctypes.c_size_t(wsb), None)
    else:
        wsb = L.dgs_spmm_csr_workspace_bytes(op, i64(M), i64(N), i64(nnz))
        ws = _ws(wsb) if wsb else None
        rc = L.dgs_spmm_csr_f32(op, i64(M), i64(K), i64(N), i64(nnz), _p(rp), _p(col), _p(val), _p(X), _p(C), _p(E), int(algorithm),
                                _p(ws), ctypes.c_size_t(wsb), None)
    assert rc == 0, f'emu spmm rc={rc}'
    global last_ws
    last_ws = ws
    return C, E


def spmm_plan(rp, col, K, compact=True):
    L = lib()
    M, nnz = rp.size - 1, col.size
    i64 = ctypes.c_int64
    pb = L.dgs_spmm_plan_bytes(i64(M), i64(K), i64(nnz))
    wb = L.dgs_spmm_plan_workspace_bytes(i64(M), i64(K), i64(nnz))
    buf, ws = _buf(pb), _buf(wb)
    info = PlanInfo()
    rc = L.dgs_spmm_plan_build(i64(M), i64(K), i64(nnz), _p(rp), _p(col), _p(buf), ctypes.c_size_t(pb), _p(ws), ctypes.c_size_t(wb),
                               ctypes.byref(info), None)
    assert rc == 0, f'emu plan build rc={rc}'
    if not compact:
        return buf, info
    cb = L.dgs_spmm_plan_compact_bytes(ctypes.byref(info))
    small = _buf(cb)
    rc = L.dgs_spmm_plan_compact(_p(buf), ctypes.byref(info), _p(small), ctypes.c_size_t(cb), i64(nnz), None)
    assert rc == 0, f'emu plan compact rc={rc}'
    return small, info


def spmm_ex(op, rp, col, val, X, bias=None, row_scale=None, relu=False, plan=None, algorithm=0):
    """dgs_spmm_csr_ex_f32: plan / info optional, fused epilogue."""
    L = lib()
    M, nnz, (K, N) = rp.size - 1, col.size, X.shape
    C = np.full((M, N), np.nan, np.float32)
    i64 = ctypes.c_int64
    pbuf, info = plan if plan is not None else (None, None)
    if plan is not None:  # (a strict call whose class thresholds were overridden ignores the plan and wants the plan-free workspace)
        wsb = max(L.dgs_spmm_csr_plan_workspace_bytes(op, i64(M), i64(N), i64(nnz), ctypes.byref(info)),
                  L.dgs_spmm_csr_workspace_bytes(op, i64(M), i64(N), i64(nnz)))
    else:
        wsb = L.dgs_spmm_csr_workspace_bytes(op, i64(M), i64(N), i64(nnz))
    ws = _ws(wsb)
    rc = L.dgs_spmm_csr_ex_f32(op, i64(M), i64(K), i64(N), i64(nnz), _p(rp), _p(col), _p(val), _p(X), _p(C), None, int(algorithm),
                               _p(bias), _p(row_scale), int(bool(relu)), _p(pbuf), ctypes.byref(info) if info is not None else None,
                               _p(ws), ctypes.c_size_t(wsb), None)
    assert rc == 0, f'emu spmm_ex rc={rc}'
    return C


def provisional_info(rp, t1=64, tslice=256):
    L = lib()
    deg = np.diff(rp).astype(np.int64)
    info = PlanInfo()
    i64 = ctypes.c_int64
    rc = L.dgs_spmm_plan_provisional_info(i64(int(deg.sum())), i64(int((deg > t1).sum())), i64(int(deg[deg > t1].sum())),
                                          i64(int((deg > tslice).sum())), i64(int(deg[deg > tslice].sum())), ctypes.byref(info))
    assert rc == 0
    return info


def schedule(op, M, K, N, nnz):
    i64 = ctypes.c_int64
    return {0: 'small', 1: 'rows', 2: 'panel'}[lib().dgs_spmm_csr_schedule(op, i64(M), i64(K), i64(N), i64(nnz))]


def sddmm(rp, col, D1, D2, mean=False, plan=None):
    L = lib()
    M, nnz, F, K = rp.size - 1, col.size, D1.shape[1], D2.shape[0]
    out = np.full(nnz, np.nan, np.float32)
    i64 = ctypes.c_int64
    if plan is not None:
        rc = L.dgs_sddmm_csr_plan_f32(MEAN if mean else SUM, i64(M), i64(K), i64(F), i64(nnz), _p(rp), _p(col), _p(D1), _p(D2), _p(out),
                                      _p(plan[0]), ctypes.byref(plan[1]), None)
    else:
        rc = L.dgs_sddmm_csr_f32(MEAN if mean else SUM, i64(M), i64(K), i64(F), i64(nnz), _p(rp), _p(col), _p(D1), _p(D2), _p(out), None)
    assert rc == 0, f'emu sddmm rc={rc}'
    return out


def csr2csc(rp, col, val, K):
    L = lib()
    L.dgs_csr2csc_workspace_bytes.restype = ctypes.c_size_t
    M, nnz = rp.size - 1, col.size
    i64 = ctypes.c_int64
    wsb = L.dgs_csr2csc_workspace_bytes(i64(M), i64(K), i64(nnz))
    ws = _buf(wsb)
    colptr = np.full(K + 1, -1, np.int32)
    row = np.full(nnz, -1, np.int32)
    cval = np.full(nnz, np.nan, np.float32)
    perm = np.full(nnz, -1, np.int32)
    rc = L.dgs_csr2csc_i32(i64(M), i64(K), i64(nnz), _p(rp), _p(col), _p(val), _p(colptr), _p(row), _p(cval), _p(perm), _p(ws),
                           ctypes.c_size_t(wsb), None)
    assert rc == 0, f'emu csr2csc rc={rc}'
    return colptr, row, cval, perm


def spmm_acc(rp, col, val, X, C, rowmap=None, plan=None):
    """C[rowmap[r]] += row r of A . X, in place."""
    L = lib()
    M, nnz, (K, N) = rp.size - 1, col.size, X.shape
    i64 = ctypes.c_int64
    if plan is not None:
        wsb = L.dgs_spmm_csr_plan_workspace_bytes(SUM, i64(M), i64(N), i64(nnz), ctypes.byref(plan[1]))
    else:
        wsb = L.dgs_spmm_csr_workspace_bytes(SUM, i64(M), i64(N), i64(nnz))
    ws = _ws(wsb)
    rc = L.dgs_spmm_csr_acc_f32(i64(M), i64(K), i64(N), i64(nnz), _p(rp), _p(col), _p(val), _p(X), _p(C), _p(rowmap),
                                _p(plan[0]) if plan is not None else None, ctypes.byref(plan[1]) if plan is not None else None,
                                _p(ws), ctypes.c_size_t(wsb), None)
    assert rc == 0, f'emu spmm_acc rc={rc}'
    return C


def spmm_acc_max(rp, col, val, X, C, Ei, rowmap, col_off, n_local, h_lo, plan=None):
    L = lib()
    M, nnz, (K, N) = rp.size - 1, col.size, X.shape
    i64, i32 = ctypes.c_int64, ctypes.c_int32
    if plan is not None:
        wsb = L.dgs_spmm_csr_plan_workspace_bytes(MAX, i64(M), i64(N), i64(nnz), ctypes.byref(plan[1]))
    else:
        wsb = L.dgs_spmm_csr_workspace_bytes(MAX, i64(M), i64(N), i64(nnz))
    ws = _ws(wsb)
    rc = L.dgs_spmm_csr_acc_max_f32(i64(M), i64(K), i64(N), i64(nnz), _p(rp), _p(col), _p(val), _p(X), _p(C), _p(Ei), _p(rowmap),
                                    i32(col_off), i32(n_local), i32(h_lo), _p(plan[0]) if plan is not None else None,
                                    ctypes.byref(plan[1]) if plan is not None else None, _p(ws), ctypes.c_size_t(wsb), None)
    assert rc == 0, f'emu spmm_acc_max rc={rc}'


def spmm_acc_min(rp, col, val, X, C, Ei, rowmap, col_off, precedes, plan=None):
    L = lib()
    M, nnz, (K, N) = rp.size - 1, col.size, X.shape
    i64, i32 = ctypes.c_int64, ctypes.c_int32
    if plan is not None:
        wsb = L.dgs_spmm_csr_plan_workspace_bytes(MIN, i64(M), i64(N), i64(nnz), ctypes.byref(plan[1]))
    else:
        wsb = L.dgs_spmm_csr_workspace_bytes(MIN, i64(M), i64(N), i64(nnz))
    ws = _ws(wsb)
    rc = L.dgs_spmm_csr_acc_min_f32(i64(M), i64(K), i64(N), i64(nnz), _p(rp), _p(col), _p(val), _p(X), _p(C), _p(Ei), _p(rowmap),
                                    i32(col_off), i32(1 if precedes else 0), _p(plan[0]) if plan is not None else None,
                                    ctypes.byref(plan[1]) if plan is not None else None, _p(ws), ctypes.c_size_t(wsb), None)
    assert rc == 0, f'emu spmm_acc_min rc={rc}'


def spmm_acc_min_around(rp, col, val, X, C, Ei, rowmap, col_off, virt_lo, virt_n, plan=None):
    """dgs_spmm_csr_acc_min_around_f32: X holds the real dense rows only (K - virt_n of them)."""
    L = lib()
    M, nnz, (Kb, N) = rp.size - 1, col.size, X.shape
    i64, i32 = ctypes.c_int64, ctypes.c_int32
    if plan is not None:
        wsb = L.dgs_spmm_csr_plan_workspace_bytes(MIN, i64(M), i64(N), i64(nnz), ctypes.byref(plan[1]))
    else:
        wsb = L.dgs_spmm_csr_workspace_bytes(MIN, i64(M), i64(N), i64(nnz))
    ws = _ws(wsb)
    rc = L.dgs_spmm_csr_acc_min_around_f32(i64(M), i64(Kb + virt_n), i64(N), i64(nnz), _p(rp), _p(col), _p(val), _p(X), _p(C),
                                           _p(Ei), _p(rowmap), i32(col_off), i32(virt_lo), i32(virt_n),
                                           _p(plan[0]) if plan is not None else None,
                                           ctypes.byref(plan[1]) if plan is not None else None, _p(ws), ctypes.c_size_t(wsb), None)
    assert rc == 0, f'emu spmm_acc_min_around rc={rc}'
