"""``Storage`` -- the CSR arrays of a sparse matrix plus its CSC view.

Behavioural mirror of the reference's ``dgsparse/storage.py:6-174`` (constructor arguments, what is asserted, which
accessor raises ``ValueError`` when its array is absent, ``Storage.empty()``, eager CSR->CSC conversion), written
independently.  Two intended differences:

* the CSR->CSC permutation comes from the integer HIP ``csr2csc`` and is exact for any nnz -- the reference pushes
  ``arange(nnz)`` through cuSPARSE as float32 *values* (storage.py:164-169), which is exact only below 2**24 entries;
* rectangular matrices work: the CSC view has ``sparse_sizes[1] = col.max() + 1`` columns (the reference passes
  ``n, n`` to cuSPARSE and is square-only).

Next to the CSC view the Storage keeps, with the same lifetime: the locality plans of the forward (CSR) and backward
(CSC) SpMM (csrc/spmm_plan.hip) and the edge values in CSC order.  The reference's operator has no per-matrix setup at
all (dgsparse/spmm.py:5-28), so a plan must never cost a caller that uses a matrix once: it is built at the
(DGS_PLAN_AFTER + 1)-th use, queued on the caller's stream without any host synchronisation (see _SharedPlan), and
Storages over the same (rowptr, col) buffers share one plan.

Reproducibility.  The plan-free and the planned schedule fold rows of 65 .. 16384 nnz with different (fixed) trees,
so sum / mean of such rows can differ in the last bits between them (max / min, rows up to 64 nnz and the hub rows above 16384
nnz are bit-identical on both).  The switch between the two is a pure function of the NUMBER of uses of the matrix - use
DGS_PLAN_AFTER + 1 is the first planned one, whatever the timing (nothing is polled: the first planned use waits for four
sums over the row lengths queued one use earlier, a few microseconds of GPU work) - so two runs of the same program agree
bit for bit.  Callers that need one schedule for the whole life of a matrix have three switches:
``torch.use_deterministic_algorithms(True)`` or ``DGS_REPRODUCIBLE=1`` (a Storage then never changes schedule on its own:
plan-free until ``storage.spmm_plan(which, n_feat, wait=True)`` is called, planned from then on), ``DGS_PLAN_AFTER=0`` (planned
from the first use) and ``DGS_PLAN=0`` (never planned).
"""
import os
import weakref
from typing import Optional

import torch

from . import _capi

_INDEX = torch.int32


def _reproducible() -> bool:
    """One schedule per matrix for its whole life: no plan is started behind the caller's back."""
    return os.environ.get('DGS_REPRODUCIBLE', '0') == '1' or torch.are_deterministic_algorithms_enabled()


def _plan_after() -> int:
    """Uses of a matrix that go plan-free before its plan is built.  Measured on the headline graph (2^20 rows, 16 M
    nnz): the build is ~1.5 ms of small launches and a planned call saves ~0.1 ms, i.e. a blocking build pays off after
    ~14 calls; queued in stream order without the synchronisation it costs the GPU 1.5 ms once and the host nothing, so 3
    calls are enough to tell a matrix that is reused (training epochs, layers sharing an adjacency) from a one-shot one
    (sampled mini-batches)."""
    try:
        return max(0, int(os.environ.get('DGS_PLAN_AFTER', '3')))
    except ValueError:
        return 3


class _SharedPlan:
    """Plan state of ONE (pointer array, index array) pair, shared by every Storage built over the same buffers.  Holds
    the arrays, so their memory cannot be recycled under a live key.

    Life of a plan: the DGS_PLAN_AFTER-th use (the last plan-free one) queues four sums over the row lengths and their copy
    to pinned memory; the next use waits for that copy (long done), queues the build on the CALLER's stream and from that very
    call on hands out the build buffer with PROVISIONAL counts (upper bounds from those sums: the kernels read the real counts
    from the device header, the host only sizes grids and the workspace with them); once the build's event has completed, a
    later use swaps in the compacted plan with the real counts and drops the worst-case build buffer (same tables, same bits).
    Which use is the first planned one depends on the use count alone, never on what has or has not completed yet."""
    __slots__ = ('ptr', 'idx', 'K', 'prefix', 'calls', 'ready', 'prov', 'stats', '__weakref__')

    def __init__(self, ptr, idx, K, prefix=None):
        self.ptr, self.idx, self.K = ptr, idx, K
        self.prefix = prefix  # pointer array of the OTHER view (colptr for the CSR plan): the column histogram, for free
        self.calls = 0
        self.ready = None   # (compact plan buffer, plan info) once the build has been seen complete
        self.prov = None    # (build buffer, provisional info, pinned header copy, event, stream) in between
        self.stats = None   # (pinned sums over the row lengths, event): what the provisional counts need; one copy for all sharers

    def get(self, wait=False):
        if self.ready is not None:
            return self.ready
        cur = torch.cuda.current_stream(self.ptr.device)
        if self.prov is None:
            if self.stats is None:
                self.stats = _length_stats(self.ptr)
            host, ev = self.stats
            ev.synchronize()  # four sums queued one use ago (or just now, DGS_PLAN_AFTER=0 / wait=True): never a poll
            buf, hdr = torch.ops.dgsparse_spmm.spmm_plan_start(self.ptr, self.idx, self.K, self.prefix)  # queued on `cur`
            done = torch.cuda.Event()
            done.record(cur)
            info = _capi.plan_provisional_info(self.idx.numel(), *host.tolist())
            self.prov = (buf, info, hdr, done, cur)
        buf, info, hdr, done, stream = self.prov
        if wait:
            done.synchronize()
        if done.query():
            if stream != cur:
                buf.record_stream(cur)  # compacted on this stream, allocated on the build's
            self.ready = tuple(torch.ops.dgsparse_spmm.spmm_plan_finish(buf, hdr, self.idx.numel()))
            self.prov = None
            return self.ready
        # in stream order behind the build the buffer is valid NOW, with provisional counts; other streams wait for `ready`
        return (buf, info) if stream == cur else (None, None)


_SHARED_PLANS = weakref.WeakValueDictionary()


_STATS_STREAMS = {}


def _length_stats(ptr: torch.Tensor):
    """Queues the four sums over the row lengths the provisional plan counts need (rows longer than t1 / tslice: how many,
    how many nnz) and their copy to pinned memory; returns (pinned int64[4], event).  They run on a side stream behind what
    the caller's stream holds NOW, so the event completes a few microseconds of GPU work later however long the caller's
    queue grows in the meantime - the use that waits for it (``_SharedPlan.get``) does not wait for the caller's own kernels."""
    dev = ptr.device
    cur = torch.cuda.current_stream(dev)
    side = _STATS_STREAMS.get(dev.index)
    if side is None:
        side = _STATS_STREAMS[dev.index] = torch.cuda.Stream(device=dev)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        t1, ts = _capi.plan_thresholds()
        deg = (ptr[1:] - ptr[:-1]).long()
        m1, m2 = deg > t1, deg > ts
        dev_stats = torch.stack([m1.sum(), (deg * m1).sum(), m2.sum(), (deg * m2).sum()])
        host = torch.empty(4, dtype=torch.int64, pin_memory=True)
        host.copy_(dev_stats, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(side)
    ptr.record_stream(side)
    return host, ev


def _index_array(t: torch.Tensor, like: torch.Tensor, numel: Optional[int] = None, dtypes=(_INDEX,)) -> torch.Tensor:
    """Asserts the reference's invariants for an index array (dtype, 1-D, same device, length) -> contiguous."""
    assert t.dtype in dtypes
    assert t.dim() == 1
    assert t.device == like.device
    if numel is not None:
        assert t.numel() == numel
    return t.contiguous()


class Storage(object):
    # attribute names are part of the surface: the reference's tests and layers reach into them
    _row: Optional[torch.Tensor]
    _rowptr: Optional[torch.Tensor]
    _col: Optional[torch.Tensor]
    _values: Optional[torch.Tensor]
    _colptr: Optional[torch.Tensor]
    _csr2csc: Optional[torch.Tensor]
    _csc2csr: Optional[torch.Tensor]
    _colcount: Optional[torch.Tensor]

    def __init__(self, row=None, rowptr=None, col=None, values=None, colptr=None, csr2csc=None, csc2csr=None,
                 colcount=None):
        assert col is not None and (rowptr is not None or row is not None)
        assert col.dtype == _INDEX and col.dim() == 1
        col = col.contiguous()
        nnz = col.numel()

        if rowptr is not None:
            n_rows = rowptr.numel() - 1
        else:
            n_rows = int(row.max()) + 1 if row.numel() else 0

        if row is not None:
            row = _index_array(row, col, nnz)
            # COO rows that define the matrix must already be in CSR order (the reference silently builds a wrong
            # matrix otherwise)
            if rowptr is None and nnz > 1 and not bool((row[1:] >= row[:-1]).all()):
                raise ValueError('dgsparse: COO row indices must be sorted (CSR order)')
        if rowptr is not None:
            rowptr = _index_array(rowptr, col, n_rows + 1)
        else:  # COO rows (sorted, as CSR order requires) -> row pointer
            rowptr = torch.zeros(n_rows + 1, dtype=_INDEX, device=col.device)
            if nnz:
                rowptr[1:] = torch.cumsum(torch.bincount(row.long(), minlength=n_rows), 0)
        # one device sync per construction, like the reference (its col.max()); the longest row rides along: callers of the
        # kernels that know it keep the launches without the hub role (DGS_ALG_NO_HUB_ROWS, hub_hints below)
        if nnz:
            cmax, self._max_row_len = torch.stack([col.max(), (rowptr[1:] - rowptr[:-1]).max()]).tolist()
            n_cols = cmax + 1
        else:
            n_cols, self._max_row_len = 0, 0
        self.sparse_sizes = (n_rows, n_cols)
        self.nnz = nnz

        if values is None:  # unit weights when the caller has none (reference: torch.ones)
            values = torch.ones(nnz, dtype=torch.float32, device=col.device)
        else:
            assert values.device == col.device and values.size(0) == nnz
            values = values.contiguous()

        # optional pre-computed CSC pieces (the reference wants int64 here and narrows later; both are accepted)
        if colptr is not None:
            colptr = _index_array(colptr, col, n_cols + 1, (torch.int64, _INDEX)).to(_INDEX)
        if csr2csc is not None:
            csr2csc = _index_array(csr2csc, col, nnz, (torch.int64, _INDEX)).to(_INDEX)
        if colcount is not None:
            colcount = _index_array(colcount, col, n_cols, (torch.int64,))

        self._row, self._rowptr, self._col, self._values = row, rowptr, col, values
        self._colptr, self._csr2csc, self._csc2csr, self._colcount = colptr, csr2csc, csc2csr, colcount
        self._csc_row = None   # row index of every CSC slot (what the backward SpMM over (colptr, row) needs)
        self._plans = {}       # 'csr' / 'csc' -> _SharedPlan (shared with every Storage over the same buffers)
        self._sched = {}       # ('csr' | 'csc', feature width) -> does that shape take the planned schedule?
        self._tvalues = None   # (weakref to values, version, values in CSC order)
        self._max_col_len = None  # longest column: int once known, (pinned tensor, event) while its copy is in flight
        self._hints = None     # (hub threshold, hint bits) once both maxima are known
        self.csr2csc_convert()
        if nnz and col.is_cuda:
            # the longest COLUMN (for the backward's transposed product) without a second sync: queued behind the transpose,
            # read when its event has completed; until then the backward simply goes without the hint
            _capi.ensure_hub_selftest(col.device)
            host = torch.empty(1, dtype=torch.int32, pin_memory=True)
            host.copy_((self._colptr[1:] - self._colptr[:-1]).max().reshape(1), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(col.device))
            self._max_col_len = (host, ev)
        elif nnz:
            self._max_col_len = int((self._colptr[1:] - self._colptr[:-1]).max())

    @classmethod
    def empty(cls):
        """A 0 x 0 matrix on the CPU (reference storage.py:102-116)."""
        nothing = torch.tensor([], dtype=_INDEX)
        return cls(row=nothing, rowptr=None, col=nothing.clone())

    # ---- accessors: the array, or ValueError when it does not exist (reference storage.py:118-157) ----------------
    def _present(self, name: str) -> torch.Tensor:
        t = getattr(self, name)
        if t is None:
            raise ValueError
        return t

    def row(self) -> torch.Tensor:
        return self._present('_row')

    def rowptr(self) -> torch.Tensor:
        return self._present('_rowptr')

    def col(self) -> torch.Tensor:
        return self._present('_col')

    def colptr(self) -> torch.Tensor:
        return self._present('_colptr')

    def values(self) -> torch.Tensor:
        return self._present('_values')

    def csr2csc(self) -> torch.Tensor:
        return self._present('_csr2csc')

    def csc_row(self) -> torch.Tensor:
        """Row index of every entry in CSC order.  The reference keeps these in ``_row`` (storage.py:170-173), which
        collides with caller-supplied COO rows (CSR order); here they always live in their own attribute."""
        return self._present('_csc_row')

    # ---- per-matrix state of the HIP schedule, built on first use ---------------------------------------------------
    def hub_hints(self) -> int:
        """`algorithm` hint bits for sum / mean over this matrix: DGS_ALG_NO_HUB_ROWS when no row is longer than the library's
        hub threshold, DGS_ALG_NO_HUB_COLS when no column is (the backward's product runs over the transpose).  True of
        every graph the reference benchmarks (Pubmed, PPI, p2p-Gnutella31, ca-CondMat: longest row 78 .. 280 nnz), which
        then keep the plain single-launch kernel; same bits with and without the hint."""
        th = _capi._lib.dgs_spmm_hub_threshold()
        hit = self._hints
        if hit is not None and hit[0] == th:
            return hit[1]
        if th <= 0:
            return 0
        bits = _capi.ALG_NO_HUB_ROWS if self._max_row_len <= th else 0
        mc = self._max_col_len
        if isinstance(mc, tuple):
            if not mc[1].query():
                return bits  # (not cached: the column maximum is still on its way)
            mc = self._max_col_len = int(mc[0][0])
        if mc is not None:
            if mc <= th:
                bits |= _capi.ALG_NO_HUB_COLS
            self._hints = (th, bits)
        return bits

    def spmm_plan(self, which: str = 'csr', n_feat: int = 64, wait: bool = False):
        """(plan buffer, plan info) of the forward ('csr': rowptr/col) or backward ('csc': colptr/csc_row) SpMM, or
        (None, None) when there is none (yet): shapes that do not take the planned schedule at this feature width
        (small inputs, dense graphs), DGS_PLAN=0, a matrix seen fewer than DGS_PLAN_AFTER + 1 times, or a stream capture
        in progress before the plan is ready (nothing may be built or polled there).  From the build on the answer is
        the build buffer with provisional counts, later the compact plan.  ``wait=True`` builds now and blocks until the
        compact plan is there (tests, benchmarks)."""
        if not (self.nnz and self._col.is_cuda) or os.environ.get('DGS_PLAN', '1') == '0':
            return (None, None)
        if which == 'csr':
            ptr, idx, M, K, other = self._rowptr, self._col, self.sparse_sizes[0], self.sparse_sizes[1], self._colptr
        else:
            ptr, idx, M, K, other = self._colptr, self._csc_row, self.sparse_sizes[1], self.sparse_sizes[0], self._rowptr
        rows = self._sched.get((which, n_feat))
        if rows is None:  # the schedule depends on the feature width: decided per width, not once per Storage
            rows = self._sched[(which, n_feat)] = M > 0 and _capi.spmm_schedule(_capi.SUM, M, K, n_feat, self.nnz) == 'rows'
        if not rows:
            return (None, None)
        sp = self._plans.get(which)
        if sp is None or sp.ptr is not ptr or sp.idx is not idx:
            key = (ptr.data_ptr(), idx.data_ptr(), ptr.numel(), idx.numel(), ptr._version, idx._version, K)
            sp = _SHARED_PLANS.get(key)
            if sp is None:
                sp = _SHARED_PLANS[key] = _SharedPlan(ptr, idx, K, other if other.numel() == K + 1 else None)
            self._plans[which] = sp
        if sp.ready is not None:
            return sp.ready
        if torch.cuda.is_current_stream_capturing():
            return (None, None)
        if sp.prov is None:
            if not wait:
                if _reproducible():
                    return (None, None)  # plan-free until the caller asks for the plan (wait=True): one schedule per matrix
                sp.calls += 1
                after = _plan_after()
                if sp.calls <= after:
                    # a matrix used once (sampled mini-batches) pays nothing; its SECOND use (or the last plan-free one) queues
                    # the row-length sums the build needs on a side stream (they used to be queued at construction,
                    # for every Storage, used or not)
                    if (sp.calls == 2 or sp.calls == after) and sp.stats is None:
                        sp.stats = _length_stats(ptr)
                    return (None, None)
        return sp.get(wait)

    def csc_values(self) -> torch.Tensor:
        """Edge values in CSC order (``values[csr2csc]``), recomputed only when ``values`` was replaced or updated in
        place (same tensor object by weak reference + its version counter: autograd's own saved-tensor rule)."""
        v = self._values
        hit = self._tvalues
        if hit is not None and hit[0]() is v and hit[1] == v._version and not torch.cuda.is_current_stream_capturing():
            return hit[2]
        out = torch.ops.dgsparse_spmm.permute_values(v.detach(), self._csr2csc)
        if not torch.cuda.is_current_stream_capturing():
            self._tvalues = (weakref.ref(v), v._version, out)
        return out

    def csr2csc_convert(self):
        """Fills in whatever is missing of (colptr, CSC row indices, CSR->CSC permutation), once.

        The CSC row indices go to ``_csc_row`` (what the spmm operators pass as their ``row`` argument); as in the
        reference (storage.py:170-173) they also land in ``_row`` when no COO rows were given."""
        if None not in (self._csr2csc, self._colptr, self._csc_row):
            return self._csr2csc
        if None not in (self._csr2csc, self._colptr):
            # pre-computed CSC view (reference storage.py:160-161 skips the conversion): the CSC row indices follow from
            # rowptr and the permutation, whatever the caller's `row` array holds
            counts = (self._rowptr[1:] - self._rowptr[:-1]).long()
            coo = torch.repeat_interleave(torch.arange(counts.numel(), device=counts.device), counts)
            self._csc_row = coo[self._csr2csc.long()].to(_INDEX)
            if self._row is None:
                self._row = self._csc_row
            return self._csr2csc
        if self.nnz == 0:
            dev = self._col.device
            colptr = torch.zeros(self.sparse_sizes[1] + 1, dtype=_INDEX, device=dev)
            csc_row = perm = torch.zeros(0, dtype=_INDEX, device=dev)
        else:
            colptr, csc_row, _, perm = _capi.csr2csc(self._rowptr, self._col, None, self.sparse_sizes[1], want_perm=True)
        self._csc_row = csc_row
        if self._row is None:
            self._row = csc_row
        if self._colptr is None:
            self._colptr = colptr
        self._csr2csc = perm
        return perm
