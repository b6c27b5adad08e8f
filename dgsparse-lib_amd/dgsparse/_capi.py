"""ctypes binding of the C-ABI library ``libdgsparse_hip.so`` (declared in include/dgsparse_hip.h).

This is the ONLY compute back end of the package: there is no CPU or eager-PyTorch fallback.  Importing
fails loudly when the library is missing, and every call on a non-GPU tensor raises.  PyTorch is used for
device memory, the current HIP stream and ``torch.distributed`` only.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DGS_LIB_PATH') or os.path.join(_HERE, 'libdgsparse_hip.so')  # (override: experiment builds)

SUM, MAX, MIN, MEAN = 0, 1, 2, 3  # include/gspmm.h:13 in the reference
ALG_SHARED_GPU = 0x100  # `algorithm` hint bit: the GPU is shared with concurrently running kernels
ALG_STRICT_SUM = 0x200  # sum / mean: one sequential fmaf chain per (row, feature) in CSR order, whatever the row length
ALG_STRICT_NOFMA = 0x400  # ... with the product rounded before the add (= the reference's host loop under g++)
ALG_NO_HUB_ROWS = 0x800  # sum / mean: the caller knows that no row is longer than hub_threshold() (kernels without the hub role)
ALG_NO_HUB_COLS = 0x1000  # ... no column is (the torch binding hands it to the backward's transposed product as NO_HUB_ROWS)

if not os.path.exists(LIB_PATH):  # mirrors dgsparse/__init__.py:25 in the reference (ImportError, no fallback)
    raise ImportError(f"Could not find the HIP kernel library '{LIB_PATH}'. Build it with "
                      f"`make -C dgsparse-lib_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`).")

_lib = ctypes.CDLL(LIB_PATH)

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_sz = ctypes.c_size_t

_lib.dgs_version.restype = _int
_lib.dgs_arch.restype = ctypes.c_char_p
_lib.dgs_strerror.restype = ctypes.c_char_p
_lib.dgs_strerror.argtypes = [_int]
_lib.dgs_spmm_hub_threshold.restype = _int
_lib.dgs_spmm_hub_threshold.argtypes = []
_lib.dgs_spmm_hub_gate.restype = _int
_lib.dgs_spmm_hub_gate.argtypes = []
_lib.dgs_spmm_hub_gate_assume.restype = _int
_lib.dgs_spmm_hub_gate_assume.argtypes = [_int]
_lib.dgs_spmm_fold_gate.restype = _int
_lib.dgs_spmm_fold_gate.argtypes = []
_lib.dgs_spmm_hub_selftest_bytes.restype = _sz
_lib.dgs_spmm_hub_selftest_bytes.argtypes = []
_lib.dgs_spmm_hub_selftest.restype = _int
_lib.dgs_spmm_hub_selftest.argtypes = [_vp, _sz, _vp]
_lib.dgs_spmm_fold_selftest.restype = _int
_lib.dgs_spmm_fold_selftest.argtypes = [_vp, _sz, _int, _int, _vp]
_lib.dgs_spmm_selftest_families.restype = _int
_lib.dgs_spmm_selftest_families.argtypes = []
_lib.dgs_spmm_selftest_hub_shapes.restype = _int
_lib.dgs_spmm_selftest_hub_shapes.argtypes = []
_lib.dgs_spmm_selftest_detail.restype = _int
_lib.dgs_spmm_selftest_detail.argtypes = [_vp, _int]
_lib.dgs_reload_tuning.restype = None
_lib.dgs_reload_tuning.argtypes = []
_lib.dgs_spmm_csr_workspace_bytes.restype = _sz
_lib.dgs_spmm_csr_workspace_bytes.argtypes = [_int, _i64, _i64, _i64]
_lib.dgs_spmm_csr_schedule.restype = _int
_lib.dgs_spmm_csr_schedule.argtypes = [_int, _i64, _i64, _i64, _i64]
_lib.dgs_spmm_csr_f32.restype = _int
_lib.dgs_spmm_csr_f32.argtypes = [_int, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _sz, _vp]


class PlanInfo(ctypes.Structure):
    """dgsSpmmPlanInfo (include/dgsparse_hip.h)."""
    _fields_ = [('n_units', ctypes.c_int32), ('n_long', ctypes.c_int32), ('n_pslots', ctypes.c_int32),
                ('n_hub', ctypes.c_int32), ('tslice', ctypes.c_int32), ('xcd_start', ctypes.c_int32 * 9),
                ('off_long', ctypes.c_int32), ('off_hub', ctypes.c_int32)]


_lib.dgs_spmm_plan_bytes.restype = _sz
_lib.dgs_spmm_plan_bytes.argtypes = [_i64, _i64, _i64]
_lib.dgs_spmm_plan_workspace_bytes.restype = _sz
_lib.dgs_spmm_plan_workspace_bytes.argtypes = [_i64, _i64, _i64]
_lib.dgs_spmm_plan_build.restype = _int
_lib.dgs_spmm_plan_build.argtypes = [_i64, _i64, _i64, _vp, _vp, _vp, _sz, _vp, _sz, ctypes.POINTER(PlanInfo), _vp]
_lib.dgs_spmm_plan_info_from_header.restype = _int
_lib.dgs_spmm_plan_info_from_header.argtypes = [_vp, _sz, ctypes.POINTER(PlanInfo)]
_lib.dgs_spmm_plan_thresholds.restype = None
_lib.dgs_spmm_plan_thresholds.argtypes = [ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
_lib.dgs_spmm_plan_provisional_info.restype = _int
_lib.dgs_spmm_plan_provisional_info.argtypes = [_i64, _i64, _i64, _i64, _i64, ctypes.POINTER(PlanInfo)]
_lib.dgs_spmm_plan_compact_bytes.restype = _sz
_lib.dgs_spmm_plan_compact_bytes.argtypes = [ctypes.POINTER(PlanInfo)]
_lib.dgs_spmm_plan_compact.restype = _int
_lib.dgs_spmm_plan_compact.argtypes = [_vp, ctypes.POINTER(PlanInfo), _vp, _sz, _i64, _vp]
_lib.dgs_spmm_csr_plan_workspace_bytes.restype = _sz
_lib.dgs_spmm_csr_plan_workspace_bytes.argtypes = [_int, _i64, _i64, _i64, ctypes.POINTER(PlanInfo)]
_lib.dgs_spmm_csr_ex_f32.restype = _int
_lib.dgs_spmm_csr_ex_f32.argtypes = [_int, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _int, _vp,
                                     ctypes.POINTER(PlanInfo), _vp, _sz, _vp]
_lib.dgs_spmm_csr_plan_f32.restype = _int
_lib.dgs_spmm_csr_plan_f32.argtypes = [_int, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                       ctypes.POINTER(PlanInfo), _vp, _sz, _vp]
_lib.dgs_spmm_csr_acc_f32.restype = _int
_lib.dgs_spmm_csr_acc_f32.argtypes = [_i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(PlanInfo), _vp, _sz,
                                      _vp]
_lib.dgs_spmm_csr_acc_max_f32.restype = _int
_lib.dgs_spmm_csr_acc_max_f32.argtypes = [_i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _vp,
                                          ctypes.POINTER(PlanInfo), _vp, _sz, _vp]
_lib.dgs_spmm_csr_acc_min_f32.restype = _int
_lib.dgs_spmm_csr_acc_min_f32.argtypes = [_i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _vp,
                                          ctypes.POINTER(PlanInfo), _vp, _sz, _vp]
_lib.dgs_spmm_csr_acc_min_around_f32.restype = _int
_lib.dgs_spmm_csr_acc_min_around_f32.argtypes = [_i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _vp,
                                                 ctypes.POINTER(PlanInfo), _vp, _sz, _vp]
_lib.dgs_spmm_csr_mask_f32.restype = _int
_lib.dgs_spmm_csr_mask_f32.argtypes = [_i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]
_lib.dgs_spmm_csr_mask_workspace_bytes.restype = _sz
_lib.dgs_spmm_csr_mask_workspace_bytes.argtypes = [_i64, _i64, _i64]
_lib.dgs_spmm_arg_backward_f32.restype = _int
_lib.dgs_spmm_arg_backward_f32.argtypes = [_i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
_lib.dgs_sddmm_csr_schedule.restype = _int
_lib.dgs_sddmm_csr_schedule.argtypes = [_i64, _i64, _i64, _i64, _int]
_lib.dgs_sddmm_csr_f32.restype = _int
_lib.dgs_sddmm_csr_f32.argtypes = [_int, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]
_lib.dgs_sddmm_csr_plan_f32.restype = _int
_lib.dgs_sddmm_csr_plan_f32.argtypes = [_int, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(PlanInfo), _vp]
_lib.dgs_sddmm_csr_mask_f32.restype = _int
_lib.dgs_sddmm_csr_mask_f32.argtypes = [_i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
_lib.dgs_csr2csc_workspace_bytes.restype = _sz
_lib.dgs_csr2csc_workspace_bytes.argtypes = [_i64, _i64, _i64]
_lib.dgs_csr2csc_i32.restype = _int
_lib.dgs_csr2csc_i32.argtypes = [_i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]
_lib.dgs_gspmm_csr_workspace_bytes.restype = _sz
_lib.dgs_gspmm_csr_workspace_bytes.argtypes = [_int, _int, _i64, _i64, _i64]
_lib.dgs_gspmm_csr_f32.restype = _int
_lib.dgs_gspmm_csr_f32.argtypes = [_int, _int, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]
_lib.dgs_sddmm_coo_f32.restype = _int
_lib.dgs_sddmm_coo_f32.argtypes = [_i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]
_lib.dgs_gather_rows_f32.restype = _int
_lib.dgs_gather_rows_f32.argtypes = [_i64, _i64, _vp, _vp, _vp, _vp]
_lib.dgs_relabel_i32.restype = _int
_lib.dgs_relabel_i32.argtypes = [_i64, _vp, _vp, _vp]
_lib.dgs_nonfinite_flag_f32.restype = _int
_lib.dgs_nonfinite_flag_f32.argtypes = [_i64, _vp, _vp, _vp]
_lib.dgs_spmm_min_merge_f32.restype = _int
_lib.dgs_spmm_min_merge_f32.argtypes = [_i64, _i64, _vp, _vp, _vp, _vp, ctypes.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
_lib.dgs_scatter_add_rows_f32.restype = _int
_lib.dgs_scatter_add_rows_f32.argtypes = [_i64, _i64, _vp, _vp, _vp, _vp]

EXPORTS = ['dgs_version', 'dgs_arch', 'dgs_strerror', 'dgs_reload_tuning', 'dgs_spmm_hub_threshold', 'dgs_spmm_hub_gate', 'dgs_spmm_hub_gate_assume', 'dgs_spmm_fold_gate',
           'dgs_spmm_hub_selftest_bytes', 'dgs_spmm_hub_selftest', 'dgs_spmm_fold_selftest', 'dgs_spmm_selftest_families', 'dgs_spmm_selftest_hub_shapes', 'dgs_spmm_selftest_detail', 'dgs_spmm_csr_workspace_bytes', 'dgs_spmm_csr_f32',
           'dgs_spmm_plan_bytes', 'dgs_spmm_plan_workspace_bytes', 'dgs_spmm_plan_build', 'dgs_spmm_plan_build2',
           'dgs_spmm_plan_compact_bytes', 'dgs_spmm_plan_compact', 'dgs_spmm_plan_info_from_header',
           'dgs_spmm_plan_thresholds', 'dgs_spmm_plan_provisional_info', 'dgs_spmm_csr_ex_f32', 'dgs_spmm_csr_plan_workspace_bytes',
           'dgs_spmm_csr_plan_f32', 'dgs_spmm_csr_acc_f32', 'dgs_spmm_csr_acc_max_f32', 'dgs_spmm_csr_acc_min_f32', 'dgs_spmm_csr_acc_min_around_f32',
           'dgs_spmm_csr_schedule', 'dgs_spmm_arg_backward_f32', 'dgs_sddmm_csr_schedule',
           'dgs_spmm_csr_mask_workspace_bytes', 'dgs_spmm_csr_mask_f32', 'dgs_sddmm_csr_f32', 'dgs_sddmm_csr_plan_f32', 'dgs_sddmm_csr_mask_f32', 'dgs_csr2csc_workspace_bytes',
           'dgs_csr2csc_i32', 'dgs_gather_rows_f32', 'dgs_scatter_add_rows_f32', 'dgs_relabel_i32', 'dgs_nonfinite_flag_f32', 'dgs_spmm_min_merge_f32', 'dgs_sddmm_coo_f32', 'dgs_gspmm_csr_workspace_bytes', 'dgs_gspmm_csr_f32', 'gespmmCsrSpMM',
           'spmm_cuda', 'spmm_cuda_no_edge_value', 'sddmm_cuda_csr', 'sddmm_cuda_coo', 'gespmmAlgSel',
           'csrspmm_parreduce_rowbalance', 'csrspmm_parreduce_nnzbalance', 'csrspmm_seqreduce_rowbalance',
           'csrspmm_seqreduce_nnzbalance', 'csrspmm_rowcaching_rowbalance', 'csrspmm_rowcaching_nnzbalance']


# ---- DGS_CANARY=1 (debug): every output and workspace this module allocates sits between two 4 KiB guard bands of a
# known pattern; canary_check() synchronises and verifies them.  The library writes through raw pointers into caller-sized
# buffers, so an overrun would otherwise be silent (tests/fuzz_gpu.py runs its campaigns with it).
_CANARY = os.environ.get('DGS_CANARY') == '1'
_GUARD, _PATTERN = 4096, 0xA5
_guarded = []


def _new(shape, dtype=torch.float32, device=None):
    if not _CANARY:
        return torch.empty(shape, dtype=dtype, device=device)
    shape = (shape,) if isinstance(shape, int) else tuple(shape)
    n = 1
    for d in shape:
        n *= int(d)
    nbytes = n * torch.empty((), dtype=dtype).element_size()
    flat = torch.full((_GUARD + nbytes + _GUARD,), _PATTERN, dtype=torch.uint8, device=device)
    _guarded.append((flat, nbytes))
    return flat[_GUARD:_GUARD + nbytes].view(dtype).view(shape)


def canary_check(what=''):
    """Verifies (and forgets) the guard bands of everything allocated since the last check; no-op without DGS_CANARY=1."""
    if not _guarded:
        return 0
    torch.cuda.synchronize()
    n = len(_guarded)
    for flat, nbytes in _guarded:
        lo, hi = flat[:_GUARD], flat[_GUARD + nbytes:]
        if not (bool((lo == _PATTERN).all()) and bool((hi == _PATTERN).all())):
            bad_lo = int((lo != _PATTERN).sum())
            bad_hi = int((hi != _PATTERN).sum())
            first_hi = int((hi != _PATTERN).nonzero()[0]) if bad_hi else -1
            _guarded.clear()
            raise AssertionError(f'dgsparse canary: {what}: buffer of {nbytes} bytes overrun: {bad_lo} guard bytes changed '
                                 f'below it, {bad_hi} above (first at +{first_hi})')
    _guarded.clear()
    return n


def version() -> int:
    return int(_lib.dgs_version())


def arch() -> str:
    return _lib.dgs_arch().decode()


def hub_threshold() -> int:
    """Rows longer than this many nnz are sequential chains in the default sum / mean schedule on the current device (0 = off:
    DGS_HUB_CHAIN=0, or the device has not passed the hub self-test - ensure_hub_selftest)."""
    if torch.cuda.is_available():
        ensure_hub_selftest(torch.device('cuda', torch.cuda.current_device()))
    return int(_lib.dgs_spmm_hub_threshold())


_selftested = set()  # device indices whose hub self-test has run in this process


# ---- optional verdict cache (DGS_GATE_CACHE=<directory>) -------------------------------------------------------------------------------
# The device gate is per process: every DataLoader worker and every rank of a job pays ~38 MB of scratch, some tens of milliseconds
# and a stream synchronisation for a verdict that is a property of (this library binary, the device model, the HIP runtime).  With
# DGS_GATE_CACHE set, a PASS is written to <dir>/dgs_gate_<key>.json and the next process with the same key adopts it
# (dgs_spmm_hub_gate_assume) instead of running the test.  Failures are never cached (they are re-tested and reported every time),
# a cache entry never turns the chains OFF, and an explicit DGS_HUB_CHAIN wins over everything as before.
def _gate_key(props: dict) -> str:
    import hashlib
    h = hashlib.sha256()
    with open(LIB_PATH, 'rb') as f:
        for chunk in iter(lambda: f.read(1 << 20), b''):
            h.update(chunk)
    for k in sorted(props):
        h.update(f'|{k}={props[k]}'.encode())
    return h.hexdigest()[:32]


def _gate_props(idx: int) -> dict:
    p = torch.cuda.get_device_properties(idx)
    return dict(name=p.name, arch=getattr(p, 'gcnArchName', ''), cus=p.multi_processor_count, mem=p.total_memory,
                hip=str(torch.version.hip), abi=int(_lib.dgs_version()))


def _gate_cache_read(directory: str, key: str):
    import json
    try:
        with open(os.path.join(directory, f'dgs_gate_{key}.json')) as f:
            d = json.load(f)
        return 1 if (d.get('key') == key and d.get('hub_chains') == 1) else None
    except (OSError, ValueError):
        return None


def _gate_cache_write(directory: str, key: str, props: dict) -> None:
    import json
    import tempfile
    try:
        os.makedirs(directory, exist_ok=True)
        fd, tmp = tempfile.mkstemp(dir=directory, prefix='.dgs_gate_')
        with os.fdopen(fd, 'w') as f:
            json.dump(dict(key=key, hub_chains=1, device=props, library=LIB_PATH), f)
        os.replace(tmp, os.path.join(directory, f'dgs_gate_{key}.json'))  # atomic: concurrent workers write the same content
    except OSError:
        pass  # a cache that cannot be written is no cache


def ensure_hub_selftest(dev) -> None:
    """Runs the library's device self-test of the hub chains once per device and process (include/dgsparse_hip.h, "Device
    gate": the default sum / mean chain their hub rows only on a device where that chain has been compared, bit for bit, with a
    one-thread-per-element sequential kernel - fourteen shapes: every family of hub workgroup, both schedules).  ~38 MB of scratch for some tens of milliseconds and ONE
    stream synchronisation, at the first use of the device; skipped (and retried later) while a stream capture is in progress;
    skipped for good - no scratch, no launch, no sync - in a process that pins DGS_HUB_CHAIN (and does not ask for DGS_FOLD=2: the
    in-kernel fold is off unless asked for, and only "2" leaves the decision to the device)."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx in _selftested:
        return
    if os.environ.get('DGS_HUB_CHAIN', '') != '' and os.environ.get('DGS_FOLD', '') != '2':
        _selftested.add(idx)  # the library would return at once as well (dgs_spmm_hub_selftest): spare the allocation
        return
    cache_dir, key, props = os.environ.get('DGS_GATE_CACHE', ''), None, None
    if cache_dir and os.environ.get('DGS_HUB_CHAIN', '') == '' and os.environ.get('DGS_FOLD', '') != '2':
        props = _gate_props(idx)
        key = _gate_key(props)
        if _gate_cache_read(cache_dir, key) == 1:  # an identical library passed on an identical device + runtime
            with _on_device(torch.device('cuda', idx)):
                _lib.dgs_spmm_hub_gate_assume(1)
            _selftested.add(idx)
            return
    if torch.cuda.is_current_stream_capturing():
        return
    _selftested.add(idx)
    d = torch.device('cuda', idx)
    with _on_device(d):
        nb = int(_lib.dgs_spmm_hub_selftest_bytes())
        scratch = torch.empty(nb, dtype=torch.uint8, device=d)
        rc = int(_lib.dgs_spmm_hub_selftest(_p(scratch), nb, _stream(d)))
        gate_now = int(_lib.dgs_spmm_hub_gate())  # (of device idx: still the current device here)
    if rc < 0:
        _selftested.discard(idx)
        _check(rc, 'spmm_hub_selftest')
    if rc == 1 and key is not None and gate_now == 1:
        _gate_cache_write(cache_dir, key, props)
    if fold_gate() < 0:
        import warnings
        warnings.warn(f'dgsparse: the in-kernel fold self-test FAILED on cuda:{idx} (DGS_FOLD=2): multi-unit rows are folded by the '
                      'combine launch on this device (same results).  Please report this.', RuntimeWarning)
    if rc == 0:
        import warnings
        warnings.warn(f'dgsparse: the hub-chain self-test FAILED on cuda:{idx} ({torch.cuda.get_device_name(idx)}): sum / mean '
                      'fold rows above 64 nnz with the fixed tree on this device (within 1e-5 of the sequential reference except '
                      'on rows of several 10^4 nnz; DGS_ALG_STRICT_SUM is unaffected).  Please report this.', RuntimeWarning)


def fold_selftest(dev=None, rounds: int = 3, load: bool = True, families=None):
    """dgs_spmm_fold_selftest on `dev` (default: the current device): the in-kernel fold against the combine launch, sum / max / min,
    every family of partial row, `rounds` times each, the last round under a streaming load from a second stream.  Returns
    (verdict 1 | 0, [mismatches per family]); a full run (families=None) moves dgs_spmm_fold_gate().  What `bench.py`, `smoke()` and
    the first_contact GPU tests call; a process gets it by itself only with DGS_FOLD=2."""
    d = dev if dev is not None else torch.device('cuda', torch.cuda.current_device())
    idx = d.index if d.index is not None else torch.cuda.current_device()
    d = torch.device('cuda', idx)
    flags = (1 if load else 0) | (sum(1 << (8 + int(f)) for f in families) if families is not None else 0)
    with _on_device(d):
        nb = int(_lib.dgs_spmm_hub_selftest_bytes())
        scratch = torch.empty(nb, dtype=torch.uint8, device=d)
        rc = int(_lib.dgs_spmm_fold_selftest(_p(scratch), nb, int(rounds), int(flags), _stream(d)))
    if rc < 0:
        _check(rc, 'spmm_fold_selftest')
    return rc, selftest_detail()[2:2 + int(_lib.dgs_spmm_selftest_families())]


def selftest_detail():
    """Mismatch counters of the last self-tests of this process: [0] hub, [1] fold, [2 + f] fold family f, [16 + h] hub shape h."""
    out = (ctypes.c_int32 * 64)()
    _lib.dgs_spmm_selftest_detail(out, 64)
    return list(out)


def hub_gate() -> int:
    """1 / 0 / -1: the hub self-test passed / has not run / failed on the current device."""
    return int(_lib.dgs_spmm_hub_gate())


def fold_gate() -> int:
    """1 / 0 / -1: the self-test of the in-kernel fold (partial rows folded by the last-arriving unit wave instead of a combine
    launch) passed / has not run / failed on the current device (fold_selftest runs it; the fold itself is off unless DGS_FOLD=1, or
    DGS_FOLD=2 on a device where it passed)."""
    return int(_lib.dgs_spmm_fold_gate())


def reload_tuning() -> None:
    """The library reads its DGS_* tuning overrides from the environment ONCE per process; call this after changing one."""
    _lib.dgs_reload_tuning()


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f'dgsparse: {what} failed: {_lib.dgs_strerror(rc).decode()} ({rc})')


def _need_gpu(*ts) -> torch.device:
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('dgsparse: this build has only the HIP (gfx950) back end - tensors must live on a GPU '
                               f'(got a {t.device} tensor); there is no CPU fallback')
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f'dgsparse: tensors on different devices ({dev} vs {t.device})')
    if dev is not None and dev.index not in _selftested:
        ensure_hub_selftest(dev)
    return dev


def _i32(t, name):
    if t.dtype != torch.int32 or t.dim() != 1:
        raise TypeError(f'dgsparse: {name} must be a 1-D int32 tensor (got {t.dtype}, dim {t.dim()})')
    return t if t.is_contiguous() else t.contiguous()


def _f32mat(t, name):
    if t.dtype != torch.float32 or t.dim() != 2:
        raise TypeError(f'dgsparse: {name} must be a 2-D float32 tensor (got {t.dtype}, dim {t.dim()})')
    return t if t.is_contiguous() else t.contiguous()


def _f32vec(t, name, n):
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise TypeError(f'dgsparse: {name} must be float32 (got {t.dtype})')
    t = t.contiguous().view(-1)
    if t.numel() != n:
        raise ValueError(f'dgsparse: {name} has {t.numel()} elements, expected {n}')
    return t


def _p(t):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream(dev):
    if _raw_stream is not None:  # ~0.3 us instead of ~1.5 us through torch.cuda.current_stream()
        return _raw_stream(dev.index if dev.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(dev).cuda_stream


class _on_device:
    """``with torch.cuda.device(dev)`` only when dev is not already current (the context manager costs ~5 us)."""
    __slots__ = ('ctx',)

    def __init__(self, dev):
        self.ctx = None if (dev.index is None or dev.index == torch.cuda.current_device()) else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def _pad4(t):
    """[R, N] -> [R, ceil4(N)] with zero columns appended."""
    n = t.shape[1]
    out = torch.zeros((t.shape[0], (n + 3) & ~3), dtype=t.dtype, device=t.device)
    out[:, :n] = t
    return out


class SpmmPlan:
    """A cached locality plan (csrc/spmm_plan.hip): device tables + the host-side counts.  Depends on (rowptr, col)
    only; never written by a call."""
    __slots__ = ('buf', 'info', 'M', 'K', 'nnz', 'rowptr_ptr', 'col_ptr')

    def __init__(self, buf, info, M, K, nnz, rowptr_ptr, col_ptr):
        self.buf, self.info, self.M, self.K, self.nnz = buf, info, M, K, nnz
        self.rowptr_ptr, self.col_ptr = rowptr_ptr, col_ptr

    def __repr__(self):
        i = self.info
        return (f'SpmmPlan(M={self.M}, nnz={self.nnz}, units={i.n_units}, long_rows={i.n_long}, pslots={i.n_pslots}, '
                f'tslice={i.tslice}, xcd_start={list(i.xcd_start)})')


def plan_thresholds():
    """(t1, tslice): the row lengths above which the plan makes units / cuts rows on the column grid."""
    a, b = ctypes.c_int32(0), ctypes.c_int32(0)
    _lib.dgs_spmm_plan_thresholds(ctypes.byref(a), ctypes.byref(b))
    return int(a.value), int(b.value)


def plan_provisional_info(nnz, rows_gt_t1, nnz_gt_t1, rows_gt_tslice, nnz_gt_tslice):
    """Upper bounds of a plan's counts as the 16-int32 CPU tensor the torch ops take as ``plan_info`` (usable with the
    build buffer on the stream the build was queued on, before anything has been read back)."""
    info = PlanInfo()
    _check(_lib.dgs_spmm_plan_provisional_info(int(nnz), int(rows_gt_t1), int(nnz_gt_t1), int(rows_gt_tslice),
                                               int(nnz_gt_tslice), ctypes.byref(info)), 'spmm_plan_provisional_info')
    t = torch.zeros(16, dtype=torch.int32)
    ctypes.memmove(t.data_ptr(), ctypes.byref(info), ctypes.sizeof(info))
    return t


def spmm_plan(rowptr, col, K, N=64, force=False):
    """Builds the locality plan of (rowptr, col) for a [K, N] dense operand, or returns None when the call would not
    use one (small inputs take one launch, dense graphs the column-panel sweep; DGS_PLAN=0 disables plans).  Once per
    matrix: ~10 launches + one host sync."""
    dev = _need_gpu(rowptr, col)
    rowptr = _i32(rowptr, 'rowptr')
    col = _i32(col, 'col')
    M, nnz = rowptr.numel() - 1, col.numel()
    if os.environ.get('DGS_PLAN', '1') == '0' and not force:
        return None
    if M <= 0 or nnz <= 0 or _lib.dgs_spmm_csr_schedule(SUM, M, int(K), int(N), nnz) != 1:
        return None
    with _on_device(dev):
        pb = _lib.dgs_spmm_plan_bytes(M, int(K), nnz)
        wb = _lib.dgs_spmm_plan_workspace_bytes(M, int(K), nnz)
        buf = _new(pb, dtype=torch.uint8, device=dev)
        ws = _new(wb, dtype=torch.uint8, device=dev)
        info = PlanInfo()
        _check(_lib.dgs_spmm_plan_build(M, int(K), nnz, _p(rowptr), _p(col), _p(buf), pb, _p(ws), wb,
                                        ctypes.byref(info), _stream(dev)), 'spmm_plan_build')
        # the build buffer is sized for the worst case (~2.9 B per nnz); keep a copy that is as large as the tables
        cb = _lib.dgs_spmm_plan_compact_bytes(ctypes.byref(info))
        small = _new(cb, dtype=torch.uint8, device=dev)
        _check(_lib.dgs_spmm_plan_compact(_p(buf), ctypes.byref(info), _p(small), cb, nnz, _stream(dev)), 'spmm_plan_compact')
    return SpmmPlan(small, info, M, int(K), nnz, rowptr.data_ptr(), col.data_ptr())


def spmm(reduce_op, rowptr, col, values, dense, algorithm=0, want_E=None, plan=None, bias=None, row_scale=None,
         relu=False):
    """C = reduce(A (*) dense).  Returns (C, E) with E=None unless max/min (or want_E).  ``plan``: a SpmmPlan of
    exactly these (rowptr, col) arrays (shapes that do not take the row-stream schedule ignore it).
    ``bias`` [N] / ``row_scale`` [M] / ``relu``: fused epilogue of sum / mean, C = relu(row_scale[:, None] * C + bias),
    bit-identical to the separate elementwise ops (dgs_spmm_csr_ex_f32)."""
    dev = _need_gpu(rowptr, col, values, dense)
    rowptr = _i32(rowptr, 'rowptr')
    col = _i32(col, 'col')
    dense = _f32mat(dense, 'dense')
    M, nnz, (K, N) = rowptr.numel() - 1, col.numel(), dense.shape
    if M < 0:
        raise ValueError('dgsparse: rowptr must have at least one element')
    epi = bias is not None or row_scale is not None or bool(relu)
    if epi:
        if reduce_op not in (SUM, MEAN):
            raise ValueError('dgsparse: the fused epilogue exists for sum and mean')
        if bias is not None:
            bias = _f32vec(bias, 'bias', N)
        if row_scale is not None:
            row_scale = _f32vec(row_scale, 'row_scale', M)
    if N % 4 and N > 4 and _lib.dgs_spmm_csr_schedule(int(reduce_op), M, K, (N + 3) & ~3, nnz) == 2:
        # dense graph, feature width not a multiple of 4 (e.g. 41 classes): the column-panel schedule needs 16-byte
        # lane vectors, and two small copies buy it (Reddit-shaped, N = 41: 4.9 -> 2.4 ms).  Feature columns are
        # independent chains, so the visible columns are bit-identical to an unpadded run.
        bp = None if bias is None else torch.cat([bias, bias.new_zeros(((N + 3) & ~3) - N)])
        C, E = spmm(reduce_op, rowptr, col, values, _pad4(dense), algorithm, want_E, bias=bp, row_scale=row_scale, relu=relu)
        return C[:, :N].contiguous(), (None if E is None else E[:, :N].contiguous())
    values = _f32vec(values, 'values', nnz)
    arg = reduce_op in (MAX, MIN) if want_E is None else want_E
    out = _new((M, N), dtype=torch.float32, device=dev)
    E = _new((M, N), dtype=torch.int32, device=dev) if arg else None
    if plan is not None and (plan.M != M or plan.nnz != nnz or plan.col_ptr != col.data_ptr() or
                             plan.rowptr_ptr != rowptr.data_ptr()):
        raise ValueError('dgsparse: the plan was built for other (rowptr, col) arrays')
    if epi:
        with _on_device(dev):
            planned = plan is not None and _lib.dgs_spmm_csr_schedule(int(reduce_op), M, K, N, nnz) == 1
            wsb = (_lib.dgs_spmm_csr_plan_workspace_bytes(reduce_op, M, N, nnz, ctypes.byref(plan.info)) if planned
                   else _lib.dgs_spmm_csr_workspace_bytes(reduce_op, M, N, nnz))
            ws = _new(wsb, dtype=torch.uint8, device=dev) if wsb else None
            _check(_lib.dgs_spmm_csr_ex_f32(reduce_op, M, K, N, nnz, _p(rowptr), _p(col), _p(values), _p(dense), _p(out), _p(E),
                                            int(algorithm), _p(bias), _p(row_scale), int(bool(relu)),
                                            _p(plan.buf) if planned else None, ctypes.byref(plan.info) if planned else None,
                                            _p(ws), wsb, _stream(dev)), 'spmm_ex')
        return out, E
    with _on_device(dev):
        strict = (int(algorithm) & (ALG_STRICT_SUM | ALG_STRICT_NOFMA)) and reduce_op in (SUM, MEAN)
        if plan is not None and strict and _lib.dgs_spmm_csr_schedule(int(reduce_op), M, K, N, nnz) == 1:
            # strict order over the plan's strict table (rows > 64 nnz sorted by length: no classify pass): the general entry
            wsb = _lib.dgs_spmm_csr_plan_workspace_bytes(reduce_op, M, N, nnz, ctypes.byref(plan.info))
            wsb = max(wsb, _lib.dgs_spmm_csr_workspace_bytes(reduce_op, M, N, nnz))  # (an experiment override may send it plan-free)
            ws = _new(wsb, dtype=torch.uint8, device=dev)
            _check(_lib.dgs_spmm_csr_ex_f32(reduce_op, M, K, N, nnz, _p(rowptr), _p(col), _p(values), _p(dense), _p(out), _p(E),
                                            int(algorithm), None, None, 0, _p(plan.buf), ctypes.byref(plan.info), _p(ws), wsb,
                                            _stream(dev)), 'spmm_ex (strict over the plan)')
            return out, E
        if plan is not None and not strict and _lib.dgs_spmm_csr_schedule(int(reduce_op), M, K, N, nnz) == 1:
            wsb = _lib.dgs_spmm_csr_plan_workspace_bytes(reduce_op, M, N, nnz, ctypes.byref(plan.info))
            ws = _new(wsb, dtype=torch.uint8, device=dev)
            _check(_lib.dgs_spmm_csr_plan_f32(reduce_op, M, K, N, nnz, _p(rowptr), _p(col), _p(values), _p(dense),
                                              _p(out), _p(E), _p(plan.buf), ctypes.byref(plan.info), _p(ws), wsb,
                                              _stream(dev)), 'spmm_plan')
            return out, E
        wsb = _lib.dgs_spmm_csr_workspace_bytes(reduce_op, M, N, nnz)
        ws = _new(wsb, dtype=torch.uint8, device=dev) if wsb else None
        _check(_lib.dgs_spmm_csr_f32(reduce_op, M, K, N, nnz, _p(rowptr), _p(col), _p(values), _p(dense), _p(out),
                                     _p(E), int(algorithm), _p(ws), wsb, _stream(dev)), 'spmm')
    return out, E


def spmm_acc(rowptr, col, values, dense, C, rowmap=None, plan=None):
    """C[rowmap[r], :] += sum_p values[p] * dense[col[p], :] in place (rowmap None: C[r, :] += ...).  Rows of A without
    entries leave C untouched.  ``plan``: the SpmmPlan of these (rowptr, col) arrays, optional."""
    dev = _need_gpu(rowptr, col, values, dense, C, rowmap)
    rowptr = _i32(rowptr, 'rowptr')
    col = _i32(col, 'col')
    dense = _f32mat(dense, 'dense')
    M, nnz, (K, N) = rowptr.numel() - 1, col.numel(), dense.shape
    if C.dtype != torch.float32 or C.dim() != 2 or C.shape[1] != N or not C.is_contiguous():
        raise TypeError('dgsparse: C must be a contiguous float32 [rows, N] tensor')
    if rowmap is not None:
        rowmap = _i32(rowmap, 'rowmap')
        if rowmap.numel() != M:
            raise ValueError('dgsparse: rowmap needs one entry per row of A')
    elif C.shape[0] < M:
        raise ValueError('dgsparse: C has fewer rows than A')
    values = _f32vec(values, 'values', nnz)
    if plan is not None and (plan.M != M or plan.nnz != nnz or plan.col_ptr != col.data_ptr() or
                             plan.rowptr_ptr != rowptr.data_ptr()):
        raise ValueError('dgsparse: the plan was built for other (rowptr, col) arrays')
    with _on_device(dev):
        if plan is not None:
            wsb = _lib.dgs_spmm_csr_plan_workspace_bytes(SUM, M, N, nnz, ctypes.byref(plan.info))
        else:
            wsb = _lib.dgs_spmm_csr_workspace_bytes(SUM, M, N, nnz)
        ws = _new(wsb, dtype=torch.uint8, device=dev) if wsb else None
        _check(_lib.dgs_spmm_csr_acc_f32(M, K, N, nnz, _p(rowptr), _p(col), _p(values), _p(dense), _p(C), _p(rowmap),
                                         _p(plan.buf) if plan is not None else None,
                                         ctypes.byref(plan.info) if plan is not None else None, _p(ws), wsb, _stream(dev)),
               'spmm_acc')
    return C


def spmm_acc_max(rowptr, col, values, dense, C, E, rowmap=None, col_off=0, n_local=0, h_lo=0, plan=None):
    """(C, E)[rowmap[r], :] <- better of what they hold and max over row r of A (args written as col + col_off), in place;
    ties go to the column that comes first in the order  [h_lo lower slots | n_local first-product columns | the rest]
    (include/dgsparse_hip.h: dgs_spmm_csr_acc_max_f32)."""
    dev = _need_gpu(rowptr, col, values, dense, C, E, rowmap)
    rowptr = _i32(rowptr, 'rowptr')
    col = _i32(col, 'col')
    dense = _f32mat(dense, 'dense')
    M, nnz, (K, N) = rowptr.numel() - 1, col.numel(), dense.shape
    if C.dtype != torch.float32 or C.dim() != 2 or C.shape[1] != N or not C.is_contiguous():
        raise TypeError('dgsparse: C must be a contiguous float32 [rows, N] tensor')
    if E.dtype != torch.int32 or E.shape != C.shape or not E.is_contiguous():
        raise TypeError('dgsparse: E must be a contiguous int32 tensor with the shape of C')
    if rowmap is not None:
        rowmap = _i32(rowmap, 'rowmap')
        if rowmap.numel() != M:
            raise ValueError('dgsparse: rowmap needs one entry per row of A')
    elif C.shape[0] < M:
        raise ValueError('dgsparse: C has fewer rows than A')
    values = _f32vec(values, 'values', nnz)
    if plan is not None and (plan.M != M or plan.nnz != nnz or plan.col_ptr != col.data_ptr() or
                             plan.rowptr_ptr != rowptr.data_ptr()):
        raise ValueError('dgsparse: the plan was built for other (rowptr, col) arrays')
    with _on_device(dev):
        if plan is not None:
            wsb = _lib.dgs_spmm_csr_plan_workspace_bytes(MAX, M, N, nnz, ctypes.byref(plan.info))
        else:
            wsb = _lib.dgs_spmm_csr_workspace_bytes(MAX, M, N, nnz)
        ws = _new(wsb, dtype=torch.uint8, device=dev) if wsb else None
        _check(_lib.dgs_spmm_csr_acc_max_f32(M, K, N, nnz, _p(rowptr), _p(col), _p(values), _p(dense), _p(C), _p(E),
                                             _p(rowmap), int(col_off), int(n_local), int(h_lo),
                                             _p(plan.buf) if plan is not None else None,
                                             ctypes.byref(plan.info) if plan is not None else None, _p(ws), wsb,
                                             _stream(dev)), 'spmm_acc_max')
    return C, E


def spmm_acc_min(rowptr, col, values, dense, C, E, rowmap=None, col_off=0, precedes=False, plan=None):
    """(C, E)[rowmap[r], :] <- algorithm 0's MIN step on what they hold and the min over row r of A (args written as
    col + col_off), in place; ``precedes``: this product's columns all come BEFORE the ones (C, E) cover in the row, else all
    AFTER (include/dgsparse_hip.h: dgs_spmm_csr_acc_min_f32)."""
    dev = _need_gpu(rowptr, col, values, dense, C, E, rowmap)
    rowptr = _i32(rowptr, 'rowptr')
    col = _i32(col, 'col')
    dense = _f32mat(dense, 'dense')
    M, nnz, (K, N) = rowptr.numel() - 1, col.numel(), dense.shape
    if C.dtype != torch.float32 or C.dim() != 2 or C.shape[1] != N or not C.is_contiguous():
        raise TypeError('dgsparse: C must be a contiguous float32 [rows, N] tensor')
    if E.dtype != torch.int32 or E.shape != C.shape or not E.is_contiguous():
        raise TypeError('dgsparse: E must be a contiguous int32 tensor with the shape of C')
    if rowmap is not None:
        rowmap = _i32(rowmap, 'rowmap')
        if rowmap.numel() != M:
            raise ValueError('dgsparse: rowmap needs one entry per row of A')
    elif C.shape[0] < M:
        raise ValueError('dgsparse: C has fewer rows than A')
    values = _f32vec(values, 'values', nnz)
    if plan is not None and (plan.M != M or plan.nnz != nnz or plan.col_ptr != col.data_ptr() or
                             plan.rowptr_ptr != rowptr.data_ptr()):
        raise ValueError('dgsparse: the plan was built for other (rowptr, col) arrays')
    with _on_device(dev):
        if plan is not None:
            wsb = _lib.dgs_spmm_csr_plan_workspace_bytes(MIN, M, N, nnz, ctypes.byref(plan.info))
        else:
            wsb = _lib.dgs_spmm_csr_workspace_bytes(MIN, M, N, nnz)
        ws = _new(wsb, dtype=torch.uint8, device=dev) if wsb else None
        _check(_lib.dgs_spmm_csr_acc_min_f32(M, K, N, nnz, _p(rowptr), _p(col), _p(values), _p(dense), _p(C), _p(E),
                                             _p(rowmap), int(col_off), int(bool(precedes)),
                                             _p(plan.buf) if plan is not None else None,
                                             ctypes.byref(plan.info) if plan is not None else None, _p(ws), wsb,
                                             _stream(dev)), 'spmm_acc_min')
    return C, E


def spmm_acc_min_around(rowptr, col, values, dense, C, E, rowmap, col_off, virt_lo, virt_n, plan=None):
    """(C, E)[rowmap[r], :] <- the MIN over row r of A in row order, where the entry with column ``virt_lo + rowmap[r]`` stands
    for what (C, E)[rowmap[r]] hold (its dense row is that row of C), columns below ``virt_lo`` are rows of ``dense`` and columns
    from ``virt_lo + virt_n`` on are rows of ``dense`` shifted by ``virt_n``; in place, ONE launch
    (include/dgsparse_hip.h: dgs_spmm_csr_acc_min_around_f32)."""
    dev = _need_gpu(rowptr, col, values, dense, C, E, rowmap)
    rowptr = _i32(rowptr, 'rowptr')
    col = _i32(col, 'col')
    dense = _f32mat(dense, 'dense')
    M, nnz, (Kb, N) = rowptr.numel() - 1, col.numel(), dense.shape
    if C.dtype != torch.float32 or C.dim() != 2 or C.shape[1] != N or not C.is_contiguous():
        raise TypeError('dgsparse: C must be a contiguous float32 [rows, N] tensor')
    if E.dtype != torch.int32 or E.shape != C.shape or not E.is_contiguous():
        raise TypeError('dgsparse: E must be a contiguous int32 tensor with the shape of C')
    rowmap = _i32(rowmap, 'rowmap')
    if rowmap.numel() != M:
        raise ValueError('dgsparse: rowmap needs one entry per row of A')
    if not (0 <= int(virt_lo) <= Kb) or int(virt_n) < C.shape[0] or Kb + int(virt_n) >= 2 ** 31:
        raise ValueError('dgsparse: virtual columns [virt_lo, virt_lo + virt_n) must start inside the dense rows and cover the rows of C')
    values = _f32vec(values, 'values', nnz)
    if plan is not None and (plan.M != M or plan.nnz != nnz or plan.col_ptr != col.data_ptr() or
                             plan.rowptr_ptr != rowptr.data_ptr()):
        raise ValueError('dgsparse: the plan was built for other (rowptr, col) arrays')
    with _on_device(dev):
        if plan is not None:
            wsb = _lib.dgs_spmm_csr_plan_workspace_bytes(MIN, M, N, nnz, ctypes.byref(plan.info))
        else:
            wsb = _lib.dgs_spmm_csr_workspace_bytes(MIN, M, N, nnz)
        ws = _new(wsb, dtype=torch.uint8, device=dev) if wsb else None
        _check(_lib.dgs_spmm_csr_acc_min_around_f32(M, Kb + int(virt_n), N, nnz, _p(rowptr), _p(col), _p(values), _p(dense),
                                                    _p(C), _p(E), _p(rowmap), int(col_off), int(virt_lo), int(virt_n),
                                                    _p(plan.buf) if plan is not None else None,
                                                    ctypes.byref(plan.info) if plan is not None else None, _p(ws), wsb,
                                                    _stream(dev)), 'spmm_acc_min_around')
    return C, E


SCHEDULES = ('small', 'rows', 'panel')


def spmm_schedule(reduce_op, M, K, N, nnz) -> str:
    """Which schedule ``spmm`` runs for these sizes: 'small' (one launch), 'rows' (row-stream + units) or 'panel'
    (column-panel sweep for dense graphs, csrc/spmm_panel.h).  Needs no GPU."""
    return SCHEDULES[_lib.dgs_spmm_csr_schedule(int(reduce_op), int(M), int(K), int(N), int(nnz))]


def spmm_mask(ptr, idx, values, grad, E, n_out=None):
    """out[j,:] = sum_p [E[idx[p],:]==j] * values[p] * grad[idx[p],:] over the CSC arrays (ptr, idx)."""
    dev = _need_gpu(ptr, idx, values, grad, E)
    ptr = _i32(ptr, 'ptr')
    idx = _i32(idx, 'idx')
    grad = _f32mat(grad, 'grad')
    if E.dtype != torch.int32 or E.shape != grad.shape:
        raise TypeError('dgsparse: E must be int32 with the shape of grad')
    E = E.contiguous()
    Mo, nnz, (Mi, N) = ptr.numel() - 1, idx.numel(), grad.shape
    values = _f32vec(values, 'values', nnz)
    rows = Mo if n_out is None else max(int(n_out), Mo)
    out = _new((rows, N), dtype=torch.float32, device=dev)
    if rows > Mo:
        out[Mo:].zero_()
    with _on_device(dev):
        wsb = _lib.dgs_spmm_csr_mask_workspace_bytes(Mo, N, nnz)
        ws = _new(wsb, dtype=torch.uint8, device=dev) if wsb else None
        _check(_lib.dgs_spmm_csr_mask_f32(Mo, Mi, N, nnz, _p(ptr), _p(idx), _p(values), _p(grad), _p(E), _p(out),
                                          _p(ws), wsb, _stream(dev)), 'spmm_mask')
    return out


def spmm_arg_backward(rowptr, col, values, E, grad, dense, need_dense=True, need_values=True):
    """Both gradients of spmm_max / spmm_min from the forward's arg ids in one pass (fp32 atomics):
    returns (grad_dense [K,N] or None, grad_values [nnz] or None)."""
    dev = _need_gpu(rowptr, col, values, E, grad, dense)
    rowptr = _i32(rowptr, 'rowptr')
    col = _i32(col, 'col')
    grad = _f32mat(grad, 'grad')
    dense = _f32mat(dense, 'dense')
    M, nnz, (K, N) = rowptr.numel() - 1, col.numel(), dense.shape
    if E.dtype != torch.int32 or tuple(E.shape) != (M, N) or tuple(grad.shape) != (M, N):
        raise TypeError('dgsparse: E must be int32 [M,N] and grad float32 [M,N]')
    E = E.contiguous()
    values = _f32vec(values, 'values', nnz)
    gX = _new((K, N), dtype=torch.float32, device=dev) if need_dense else None
    gW = _new(nnz, dtype=torch.float32, device=dev) if (need_values and values is not None) else None
    with _on_device(dev):
        _check(_lib.dgs_spmm_arg_backward_f32(M, K, N, nnz, _p(rowptr), _p(col), _p(values), _p(E), _p(grad), _p(dense),
                                              _p(gX), _p(gW), _stream(dev)), 'spmm_arg_backward')
    return gX, gW


def sddmm(rowptr, col, D1, D2, reduce_op=SUM, E=None, plan=None):
    """out[e] = <D1[row(e)], D2[col(e)]> (mean-scaled / arg-masked variants).  ``plan``: the SpmmPlan of exactly these
    (rowptr, col) arrays (the unmasked product then takes the fused row-block / unit schedule where it applies)."""
    dev = _need_gpu(rowptr, col, D1, D2, E)
    rowptr = _i32(rowptr, 'rowptr')
    col = _i32(col, 'col')
    D1 = _f32mat(D1, 'D1')
    D2 = _f32mat(D2, 'D2')
    M, nnz, F = rowptr.numel() - 1, col.numel(), D1.shape[1]
    if D2.shape[1] != F or D1.shape[0] < M:
        raise ValueError(f'dgsparse: sddmm shape mismatch D1 {tuple(D1.shape)} D2 {tuple(D2.shape)} rows {M}')
    if F % 4 and F > 4 and _lib.dgs_sddmm_csr_schedule(M, D2.shape[0], (F + 3) & ~3, nnz, int(E is not None)) == 2:
        # same trick as in spmm(): zero feature columns add exact zeros to every dot product
        Ep = None
        if E is not None:
            Ep = torch.full((E.shape[0], (F + 3) & ~3), -1, dtype=torch.int32, device=dev)
            Ep[:, :F] = E
        return sddmm(rowptr, col, _pad4(D1), _pad4(D2), reduce_op, Ep)
    out = _new(nnz, dtype=torch.float32, device=dev)
    if plan is not None and (plan.M != M or plan.nnz != nnz or plan.col_ptr != col.data_ptr() or
                             plan.rowptr_ptr != rowptr.data_ptr()):
        raise ValueError('dgsparse: the plan was built for other (rowptr, col) arrays')
    with _on_device(dev):
        if E is None and plan is not None:
            _check(_lib.dgs_sddmm_csr_plan_f32(reduce_op, M, D2.shape[0], F, nnz, _p(rowptr), _p(col), _p(D1), _p(D2),
                                               _p(out), _p(plan.buf), ctypes.byref(plan.info), _stream(dev)), 'sddmm_plan')
        elif E is not None:
            if E.dtype != torch.int32 or E.shape != D1.shape:
                raise TypeError('dgsparse: E must be int32 with the shape of D1')
            _check(_lib.dgs_sddmm_csr_mask_f32(M, D2.shape[0], F, nnz, _p(rowptr), _p(col), _p(D1), _p(D2),
                                               _p(E.contiguous()), _p(out), _stream(dev)), 'sddmm_mask')
        else:
            _check(_lib.dgs_sddmm_csr_f32(reduce_op, M, D2.shape[0], F, nnz, _p(rowptr), _p(col), _p(D1), _p(D2),
                                          _p(out), _stream(dev)), 'sddmm')
    return out


def gspmm(reduce_op, compute_op, rowptr, col, values, dense):
    """C = reduce_p compute(values[p], dense[col[p]]); compute_op 0 add / 1 sub (x - w) / 2 mul / 3 div (x / w)."""
    dev = _need_gpu(rowptr, col, values, dense)
    rowptr = _i32(rowptr, 'rowptr')
    col = _i32(col, 'col')
    dense = _f32mat(dense, 'dense')
    M, nnz, (K, N) = rowptr.numel() - 1, col.numel(), dense.shape
    values = _f32vec(values, 'values', nnz)
    out = _new((M, N), dtype=torch.float32, device=dev)
    with _on_device(dev):
        wsb = _lib.dgs_gspmm_csr_workspace_bytes(reduce_op, compute_op, M, N, nnz)
        ws = _new(wsb, dtype=torch.uint8, device=dev) if wsb else None
        _check(_lib.dgs_gspmm_csr_f32(reduce_op, compute_op, M, K, N, nnz, _p(rowptr), _p(col), _p(values), _p(dense),
                                      _p(out), _p(ws), wsb, _stream(dev)), 'gspmm')
    return out


def sddmm_coo(rowind, colind, D1, D2):
    """out[e] = <D1[rowind[e]], D2[colind[e]]> for COO index arrays."""
    dev = _need_gpu(rowind, colind, D1, D2)
    rowind = _i32(rowind, 'rowind')
    colind = _i32(colind, 'colind')
    D1 = _f32mat(D1, 'D1')
    D2 = _f32mat(D2, 'D2')
    if rowind.numel() != colind.numel() or D1.shape[1] != D2.shape[1]:
        raise ValueError('dgsparse: sddmm_coo shape mismatch')
    out = _new(rowind.numel(), dtype=torch.float32, device=dev)
    with _on_device(dev):
        _check(_lib.dgs_sddmm_coo_f32(D1.shape[1], rowind.numel(), _p(rowind), _p(colind), _p(D1), _p(D2), _p(out),
                                      _stream(dev)), 'sddmm_coo')
    return out


def csr2csc(rowptr, col, values, n_cols, want_perm=True):
    """Stable transpose.  Returns (colptr[n_cols+1], row[nnz], cscval[nnz] | None, perm[nnz] | None)."""
    dev = _need_gpu(rowptr, col, values)
    rowptr = _i32(rowptr, 'rowptr')
    col = _i32(col, 'col')
    M, nnz = rowptr.numel() - 1, col.numel()
    values = _f32vec(values, 'values', nnz)
    colptr = _new(n_cols + 1, dtype=torch.int32, device=dev)
    row = _new(nnz, dtype=torch.int32, device=dev)
    cscval = _new(nnz, dtype=torch.float32, device=dev) if values is not None else None
    perm = _new(nnz, dtype=torch.int32, device=dev) if want_perm else None
    with _on_device(dev):
        wsb = _lib.dgs_csr2csc_workspace_bytes(M, n_cols, nnz)
        ws = _new(wsb, dtype=torch.uint8, device=dev)
        _check(_lib.dgs_csr2csc_i32(M, n_cols, nnz, _p(rowptr), _p(col), _p(values), _p(colptr), _p(row), _p(cscval),
                                    _p(perm), _p(ws), wsb, _stream(dev)), 'csr2csc')
    return colptr, row, cscval, perm


def gather_rows(src, ids):
    dev = _need_gpu(src, ids)
    src = _f32mat(src, 'src')
    ids = _i32(ids, 'ids')
    out = _new((ids.numel(), src.shape[1]), dtype=torch.float32, device=dev)
    with _on_device(dev):
        _check(_lib.dgs_gather_rows_f32(ids.numel(), src.shape[1], _p(ids), _p(src), _p(out), _stream(dev)), 'gather')
    return out


def scatter_add_rows(dst, ids, src):
    dev = _need_gpu(dst, ids, src)
    assert dst.is_contiguous() and dst.dtype == torch.float32
    src = _f32mat(src, 'src')
    ids = _i32(ids, 'ids')
    with _on_device(dev):
        _check(_lib.dgs_scatter_add_rows_f32(ids.numel(), src.shape[1], _p(ids), _p(src), _p(dst), _stream(dev)),
               'scatter_add')
    return dst


def nonfinite_flag(x, flag):
    """flag[0] |= 1 when x holds a NaN or an infinity (flag: int32 device tensor the caller zeroed); no host sync."""
    dev = _need_gpu(x, flag)
    if x.dtype != torch.float32 or not x.is_contiguous() or flag.dtype != torch.int32 or flag.numel() < 1:
        raise TypeError('dgsparse: nonfinite_flag wants a contiguous float32 tensor and an int32 flag')
    with _on_device(dev):
        _check(_lib.dgs_nonfinite_flag_f32(x.numel(), _p(x), _p(flag), _stream(dev)), 'nonfinite_flag')
    return flag


def spmm_min_merge(rowmap, rowptr2, Ch, Eh, col_off, loc_rowptr, C, E, flag, rowptr, col, values, dense):
    """In place (C, E)[rowmap[r]] <- MIN fold of  Ch[2r] | (C, E)[rowmap[r]] | Ch[2r + 1]  in CSR order, or the sequential
    chain over the whole shard row when flag[0] != 0 (include/dgsparse_hip.h: dgs_spmm_min_merge_f32).  rowptr2 = None:
    redo-only call (Ch, Eh, loc_rowptr ignored) after the accumulating min kernels did the merge."""
    dev = _need_gpu(rowmap, rowptr2, Ch, Eh, loc_rowptr, C, E, flag, rowptr, col, values, dense)
    rowmap = _i32(rowmap, 'rowmap')
    rowptr, col = _i32(rowptr, 'rowptr'), _i32(col, 'col')
    dense = _f32mat(dense, 'dense')
    R, N = rowmap.numel(), dense.shape[1]
    checks = [('C', C, torch.float32, None), ('E', E, torch.int32, None)]
    if rowptr2 is not None:
        rowptr2, loc_rowptr = _i32(rowptr2, 'rowptr2'), _i32(loc_rowptr, 'loc_rowptr')
        if rowptr2.numel() != 2 * R + 1:
            raise ValueError('dgsparse: rowptr2 needs 2 * len(rowmap) + 1 entries')
        if loc_rowptr.numel() != C.shape[0] + 1:
            raise ValueError('dgsparse: loc_rowptr and C describe different row counts')
        checks += [('Ch', Ch, torch.float32, 2 * R), ('Eh', Eh, torch.int32, 2 * R)]
    else:
        Ch = Eh = loc_rowptr = None
    for name, t, dt, rows in checks:
        if t.dtype != dt or t.dim() != 2 or t.shape[1] != N or not t.is_contiguous() or (rows is not None and t.shape[0] != rows):
            raise TypeError(f'dgsparse: {name} must be a contiguous {dt} [rows, {N}] tensor')
    if E.shape != C.shape or rowptr.numel() != C.shape[0] + 1:
        raise ValueError('dgsparse: C, E and rowptr describe different row counts')
    if flag is None or flag.dtype != torch.int32 or flag.numel() < 1:
        raise TypeError('dgsparse: flag must be an int32 device tensor')
    values = _f32vec(values, 'values', col.numel())
    with _on_device(dev):
        _check(_lib.dgs_spmm_min_merge_f32(R, N, _p(rowmap), _p(rowptr2), _p(Ch), _p(Eh), int(col_off), _p(loc_rowptr),
                                           _p(C), _p(E), _p(flag), _p(rowptr), _p(col), _p(values), _p(dense),
                                           _stream(dev)), 'spmm_min_merge')
    return C, E


def relabel_(ids, mapping):
    """In place: ids[i] = mapping[ids[i]] where ids[i] >= 0 (int32 tensors; negative ids stay)."""
    dev = _need_gpu(ids, mapping)
    if ids.dtype != torch.int32 or mapping.dtype != torch.int32 or not ids.is_contiguous() or not mapping.is_contiguous():
        raise TypeError('dgsparse: relabel_ wants contiguous int32 tensors')
    with _on_device(dev):
        _check(_lib.dgs_relabel_i32(ids.numel(), _p(ids), _p(mapping), _stream(dev)), 'relabel')
    return ids
