"""Multi-GPU SpMM: 1-D row partition + halo feature exchange (new design; the reference is single-device only,
SURVEY.md R4 / section 8e).

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for the tests).
Rank g owns a contiguous block of rows of A and the matching rows of the dense operand B and of the output C.
A row of A may reference columns owned by other ranks; those feature rows ("halo") are fetched once per SpMM call
with ONE all-to-all-v:

    offline  HaloPlan:  unique remote column ids grouped by owner  -> recv_ids   (what I need)
                        transpose of the plan (all-to-all of counts + ids) -> send_ids (what peers need from me)
                        local CSR re-labelled into the extended index space [local rows | halo slots]
    per call pack   : gather_rows(B_loc, send_ids)            (HIP kernel dgs_gather_rows_f32)
             exchange: all_to_all_single(B_ext[Mloc:], packed, recv_splits, send_splits)   (RCCL)
             compute : sum/mean: C = spmm(A_loc, B_loc) WHILE the exchange is in flight (RCCL runs on its own
                                 stream), then C[rem_rows] += spmm(A_rem, B_halo)   [A = A_loc + A_rem by column owner;
                                 A_rem holds only the rows that have a remote entry]
                       max/min : shards with sorted rows overlap too - (C, E) of the local columns while the halo
                                 travels, then the halo part is merged in CSR order (max: accumulating kernel, ties to the
                                 smaller global column; min: the lower-rank halo entries folded in front of the local
                                 result and the higher-rank ones behind it, two accumulating launches, with a
                                 sequential redo when a NaN / inf is about) - bit-exact;
                                 unsorted rows: C, E = spmm(A_ext, B_ext) after the exchange (one pass)
    backward (DistSpMMFn): the same plan reversed - gradients of halo rows travel home by all-to-all-v and are
             scatter-added; sum / mean / max / min, w.r.t. the feature rows and the edge values.

xGMI is point-to-point (7 links per GPU), so an all-to-all uses every link at once - the right collective shape for
this fabric; there is no all-reduce anywhere.  Only feature rows travel; the graph never does.
max/min return global column ids in E (ext -> global relabel), first-occurrence ties are preserved because the
order of a row's entries is never changed.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist


@dataclass
class RowPartition:
    """Rows [row_offsets[rank], row_offsets[rank+1]) of a global M x M CSR; ``col`` holds GLOBAL column ids."""
    rank: int
    world: int
    row_offsets: List[int]
    rowptr: torch.Tensor  # int32 [Mloc+1]
    col: torch.Tensor     # int32 [nnz], global ids, CSR order inside rows
    val: Optional[torch.Tensor]  # float32 [nnz] or None

    @property
    def r0(self):
        return self.row_offsets[self.rank]

    @property
    def n_local(self):
        return self.row_offsets[self.rank + 1] - self.row_offsets[self.rank]

    @property
    def nnz(self):
        return int(self.col.numel())


def row_offsets(rowptr, world: int, balance: str = 'rows') -> List[int]:
    """Boundaries of ``world`` CONTIGUOUS row blocks.  'rows': equal row counts.  'nnz': block g ends at the first row
    whose prefix nnz reaches (g + 1) * nnz / world - on a power-law graph in a degree-correlated order equal row blocks
    give one rank the hubs (the reference's own load-balance idea, nnz-balanced work: src/ge-spmm/
    csrspmm_rowcaching.cu:121-294, applied to the partition).  The rows of B and C follow the same boundaries."""
    rowptr = torch.as_tensor(rowptr)
    M = rowptr.numel() - 1
    if balance == 'rows':
        per = (M + world - 1) // world
        return [min(M, i * per) for i in range(world + 1)]
    if balance != 'nnz':
        raise ValueError(balance)
    nnz = int(rowptr[-1])
    targets = torch.tensor([(nnz * g) // world for g in range(1, world)], dtype=rowptr.dtype, device=rowptr.device)
    cuts = torch.searchsorted(rowptr.contiguous(), targets, right=False).clamp_(0, M).tolist() if world > 1 else []
    offs = [0] + [int(c) for c in cuts] + [M]
    for i in range(1, len(offs)):  # monotone even with runs of empty rows
        offs[i] = max(offs[i], offs[i - 1])
    return offs


def partition_csr(rowptr, col, val, world: int, balance: str = 'rows') -> List[RowPartition]:
    """Split a global CSR (numpy or torch, square) into ``world`` contiguous row blocks: equal row counts
    (``balance='rows'``) or equal nnz (``'nnz'``, see row_offsets)."""
    rowptr = torch.as_tensor(rowptr)
    col = torch.as_tensor(col)
    val = None if val is None else torch.as_tensor(val)
    offs = row_offsets(rowptr, world, balance)
    parts = []
    for r in range(world):
        s, e = int(rowptr[offs[r]]), int(rowptr[offs[r + 1]])
        parts.append(RowPartition(r, world, offs, (rowptr[offs[r]:offs[r + 1] + 1] - s).to(torch.int32).contiguous(),
                                  col[s:e].to(torch.int32).contiguous(),
                                  None if val is None else val[s:e].to(torch.float32).contiguous()))
    return parts


def synthetic_partition(rank: int, world: int, n_local: int, deg: int, cols: str = 'powerlaw', locality: float = 0.8,
                        alpha: float = 2.1, dmax: int = 1 << 16, seed: int = 0, device='cpu') -> RowPartition:
    """Rank-local block of a synthetic power-law graph with ``n_local`` rows per rank (weak scaling).

    Row degrees: truncated Pareto (as bench/graphgen.py).  Each entry stays inside the rank's own row block with
    probability ``locality`` (1 - edge cut; partitioned real graphs have cuts of 10-30 %), else it goes to a
    uniformly chosen other rank.  Inside the owner's block the column follows a popularity law that depends on
    (seed, owner) only, so every rank agrees on which columns are hubs."""
    from bench import graphgen  # generators live next to bench.py; only used by benchmarks and tests
    dev = torch.device(device)
    rng = np.random.Generator(np.random.PCG64(seed * 1000 + rank))
    degs = graphgen.powerlaw_degrees(n_local, n_local * deg, alpha, min(dmax, n_local), rng)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed * 1000 + rank)
    deg_t = torch.from_numpy(degs).to(dev)
    total = int(deg_t.sum())
    row = torch.repeat_interleave(torch.arange(n_local, device=dev), deg_t)
    if world > 1:
        remote = torch.rand(total, generator=gen, device=dev) >= locality
        other = torch.randint(0, world - 1, (total,), generator=gen, device=dev)
        other = other + (other >= rank).long()
        owner = torch.where(remote, other, torch.full_like(other, rank))
    else:
        owner = torch.zeros(total, dtype=torch.long, device=dev)
    if cols == 'uniform':
        inner = torch.randint(0, n_local, (total,), generator=gen, device=dev)
    else:
        inner = torch.empty(total, dtype=torch.long, device=dev)
        u = torch.rand(total, generator=gen, device=dev, dtype=torch.float64)
        for h in range(world):  # popularity of owner h's columns: a function of (seed, h) only
            prng = np.random.Generator(np.random.PCG64(seed * 7919 + h))
            w = graphgen.powerlaw_degrees(n_local, n_local * deg, alpha, min(dmax, n_local), prng).astype(np.float64) + 0.05
            w = w[prng.permutation(n_local)]
            cdf = torch.from_numpy(np.cumsum(w) / w.sum()).to(dev)
            m = owner == h
            inner[m] = torch.searchsorted(cdf, u[m], right=True).clamp_(max=n_local - 1)
    K = world * n_local
    key = torch.unique_consecutive(torch.sort(row * K + owner * n_local + inner).values)
    row = key // K
    col = (key - row * K).to(torch.int32)
    counts = torch.bincount(row, minlength=n_local)
    rowptr = torch.zeros(n_local + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=rowptr[1:])
    vg = torch.Generator(device=dev)
    vg.manual_seed(seed * 1000 + rank + 17)
    val = torch.rand(col.numel(), generator=vg, device=dev)
    return RowPartition(rank, world, [i * n_local for i in range(world + 1)], rowptr.to(torch.int32), col, val)


class _Done:
    """Work handle of a collective that has already completed (the staged gloo path below)."""

    def wait(self):
        return True


def _staged(t: torch.Tensor, group) -> bool:
    """GPU tensors on a gloo group: gloo has no all-to-all for device memory, so the collective is staged through the host.
    Never the production path (backend nccl = RCCL moves device memory directly); it lets SEVERAL ranks share ONE GPU, which
    is how tests/test_gpu_dist.py runs the N > 1 code paths with the real HIP kernels on a single-GPU box."""
    return t.is_cuda and dist.get_backend(group) == 'gloo'


def _a2a(out: torch.Tensor, inp: torch.Tensor, out_splits: List[int], in_splits: List[int], group=None, async_op=False):
    """all_to_all_single with per-peer splits (first-dim rows); returns a work handle when async_op."""
    if _staged(out, group):
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(host, inp.cpu().contiguous(), out_splits, in_splits, group=group)
        out.copy_(host)
        return _Done() if async_op else None
    return dist.all_to_all_single(out, inp, out_splits, in_splits, group=group, async_op=async_op)


def _all_to_all_v(out: torch.Tensor, inp: torch.Tensor, out_splits: List[int], in_splits: List[int], group=None):
    """all_to_all_single with per-peer splits (first-dim rows); nccl(RCCL) and gloo both implement it."""
    _a2a(out, inp, out_splits, in_splits, group)


class HaloPlan:
    """Who needs which feature rows.  Built once per partition (collective: every rank must call it, unless
    ``standalone``: then only the receive side is built - what one rank's shard needs - and nothing is exchanged; the
    caller fills the halo rows itself.  That is how a single GPU runs ONE rank's shard of a many-rank job)."""

    def __init__(self, part: RowPartition, group=None, standalone: bool = False):
        dev = part.col.device
        world, rank = part.world, part.rank
        offs = torch.tensor(part.row_offsets, device=dev, dtype=torch.int64)
        col = part.col.long()
        owner = torch.bucketize(col, offs[1:], right=True)  # owner[i] = rank whose block contains col[i]
        is_remote = owner != rank
        rem = torch.unique(col[is_remote])  # sorted => grouped by owner (blocks are contiguous)
        rem_owner = torch.bucketize(rem, offs[1:], right=True)
        recv_splits = torch.bincount(rem_owner, minlength=world)
        # extended column ids: local -> [0, n_local); remote -> n_local + position in `rem`
        ext = torch.where(is_remote, part.n_local + torch.searchsorted(rem, col), col - part.r0)
        self.col_ext = ext.to(torch.int32).contiguous()
        self.recv_ids = rem  # global ids, ascending
        self.recv_splits = recv_splits.tolist()
        self.n_halo = int(rem.numel())
        self.ext2glob = torch.cat([torch.arange(part.r0, part.r0 + part.n_local, device=dev), rem])
        self.ext2glob32 = self.ext2glob.to(torch.int32).contiguous()
        # A = A_loc + A_rem (same rows): lets the local product run while the halo is still in flight.  A_rem is stored
        # COMPACT: only the rows that have a remote entry (rem_rows), so that the second product and the add touch
        # those rows only (with an edge cut of 20 % and a median degree of 4, four rows in ten have none)
        nl = part.n_local
        counts = (part.rowptr[1:] - part.rowptr[:-1]).long()
        rows = torch.repeat_interleave(torch.arange(nl, device=dev), counts)
        self.nnz_pos_loc = torch.nonzero(~is_remote).view(-1)  # positions of the local / remote entries in part.col
        self.nnz_pos_rem = torch.nonzero(is_remote).view(-1)

        def _sub(mask, shift, compact):
            cnt = torch.bincount(rows[mask], minlength=nl)
            keep = torch.nonzero(cnt).view(-1) if compact else None
            if compact:
                cnt = cnt[keep]
            rp = torch.zeros(cnt.numel() + 1, dtype=torch.int64, device=dev)
            rp[1:] = torch.cumsum(cnt, 0)
            return (rp.to(torch.int32), (ext[mask] - shift).to(torch.int32).contiguous(),
                    None if part.val is None else part.val[mask].contiguous()), keep

        self.loc, _ = _sub(~is_remote, 0, False)
        self.rem, keep = _sub(is_remote, nl, True)
        self.rem_rows = keep.to(torch.int32).contiguous()
        self.deg = counts.clamp(min=1).to(torch.float32)
        # for the overlapped max: halo slots owned by lower ranks (they precede the local columns in a sorted row), and
        # whether every row's columns are sorted (then "smaller global column" = "earlier in CSR order")
        self.h_lo = int(recv_splits[:rank].sum())
        inc = torch.ones(col.numel(), dtype=torch.bool, device=dev)
        if col.numel() > 1:
            inc[1:] = col[1:] >= col[:-1]
            starts = part.rowptr[1:-1].long()
            inc[starts[starts < col.numel()]] = True  # a REAL row start never breaks the order (trailing empty rows have
            # rowptr == nnz: clamping those to nnz-1 used to mask a descent inside the last non-empty row)
        self.rows_sorted = bool(inc.all())
        self._min_parts = None  # see min_parts()
        if world == 1 or standalone:  # nothing to exchange; no process group needed
            self.send_splits = [0] * world
            self.send_ids = torch.zeros(0, dtype=torch.int32, device=dev)
            return
        # transpose the plan: tell every owner which of its rows I need
        send_counts = torch.empty(world, dtype=torch.int64, device=dev)
        _all_to_all_v(send_counts, recv_splits.contiguous(), [1] * world, [1] * world, group)
        self.send_splits = send_counts.tolist()
        send_ids = torch.empty(int(send_counts.sum()), dtype=torch.int64, device=dev)
        _all_to_all_v(send_ids, rem.contiguous(), self.send_splits, self.recv_splits, group)
        self.send_ids = (send_ids - part.r0).to(torch.int32).contiguous()  # local row indices peers asked for
        assert self.send_ids.numel() == 0 or (int(self.send_ids.min()) >= 0 and int(self.send_ids.max()) < part.n_local)


    def min_parts(self):
        """For the overlapped min: MIN can only be folded in row order (csrc/spmm_impl.h AccArg), so the halo matrix is kept a
        second time as two compact matrices - the slots of lower ranks (they precede the local columns of a sorted row) and
        those of higher ranks (they follow them).  Built on the first min call, from the halo matrix alone.
        Returns [(matrix (rowptr, slots, values | None), shard rows it touches, positions of its entries in the halo
        matrix's arrays), ...] for (lower, higher)."""
        if self._min_parts is None:
            assert self.rows_sorted
            rp, slot, v = self.rem
            dev = slot.device
            R = int(self.rem_rows.numel())
            rpl = rp.long()
            rrow = torch.repeat_interleave(torch.arange(R, device=dev), rpl[1:] - rpl[:-1])
            lower = slot.long() < self.h_lo
            parts = []
            for mask in (lower, ~lower):
                cnt = torch.bincount(rrow[mask], minlength=R)
                keep = torch.nonzero(cnt).view(-1)
                rp2 = torch.zeros(keep.numel() + 1, dtype=torch.int64, device=dev)
                rp2[1:] = torch.cumsum(cnt[keep], 0)
                pos = torch.nonzero(mask).view(-1)
                parts.append(((rp2.to(torch.int32), slot[pos].contiguous(), None if v is None else v[pos].contiguous()),
                              self.rem_rows[keep].contiguous(), pos))
            self._min_parts = parts
        return self._min_parts


    def min_around(self):
        """For the overlapped min in ONE accumulating launch (round 5; include/dgsparse_hip.h
        dgs_spmm_csr_acc_min_around_f32): the halo matrix once more, in the id space [slots of lower ranks | one VIRTUAL column
        per shard row | slots of higher ranks] = the global column order of a sorted shard row, every row that also has local
        entries carrying the virtual entry (h_lo + shard row, weight 1) that stands for its local result.
        Returns ((rowptr, ids, values | None), shard rows it touches, pos) with pos[i] = position of entry i in the halo
        matrix's arrays, -1 for a virtual entry (how a caller-supplied ``val`` finds its way in)."""
        if getattr(self, '_min_around', None) is None:
            assert self.rows_sorted
            rp, slot, v = self.rem
            dev = slot.device
            R = int(self.rem_rows.numel())
            nl = int(self.loc[0].numel()) - 1
            rpl = rp.long()
            rrow = torch.repeat_interleave(torch.arange(R, device=dev), rpl[1:] - rpl[:-1])
            sl = slot.long()
            ids = torch.where(sl < self.h_lo, sl, sl + nl)
            lrp = self.loc[0].long()
            has_loc = (lrp[1:] - lrp[:-1])[self.rem_rows.long()] > 0  # per compact row
            vrow = torch.nonzero(has_loc).view(-1)
            arow = torch.cat([rrow, vrow])
            aid = torch.cat([ids, self.h_lo + self.rem_rows.long()[vrow]])
            apos = torch.cat([torch.arange(sl.numel(), device=dev), torch.full((vrow.numel(),), -1, dtype=torch.long, device=dev)])
            order = torch.argsort(arow * (self.n_halo + nl + 1) + aid)  # keys are unique: (row, id)
            aid, apos = aid[order], apos[order]
            cnt = torch.bincount(arow, minlength=R)
            rp2 = torch.zeros(R + 1, dtype=torch.int64, device=dev)
            rp2[1:] = torch.cumsum(cnt, 0)
            vals = None
            if v is not None:
                vals = torch.where(apos >= 0, v[apos.clamp(min=0)], torch.ones((), dtype=v.dtype, device=dev)).contiguous()
            self._min_around = ((rp2.to(torch.int32), aid.to(torch.int32).contiguous(), vals), self.rem_rows, apos)
        return self._min_around


class _HipOps:
    """The product compute path: the HIP kernels through the C ABI (with cached locality plans per matrix)."""

    def __init__(self):
        from . import _capi
        self._c = _capi
        self._plans = {}

    def _plan(self, rowptr, col, K, N):
        key = (rowptr.data_ptr(), col.data_ptr(), int(col.numel()))
        if key not in self._plans:
            self._plans[key] = self._c.spmm_plan(rowptr, col, K, N) if col.numel() else None
        return self._plans[key]

    def spmm(self, op, rowptr, col, val, B, shared_gpu=False):
        """shared_gpu: a collective is in flight on another stream: keep off schedules that need every CU."""
        return self._c.spmm(op, rowptr, col, val, B, algorithm=self._c.ALG_SHARED_GPU if shared_gpu else 0,
                            plan=self._plan(rowptr, col, B.shape[0], B.shape[1]))

    def spmm_acc(self, rowptr, col, val, B, C, rowmap):
        """C[rowmap[r]] += row r of A.B, in place (dgs_spmm_csr_acc_f32)."""
        return self._c.spmm_acc(rowptr, col, val, B, C, rowmap, plan=self._plan(rowptr, col, B.shape[0], B.shape[1]))

    def spmm_acc_max(self, rowptr, col, val, B, C, E, rowmap, col_off, n_local, h_lo):
        """(C, E)[rowmap[r]] <- better of the old pair and the max over row r (dgs_spmm_csr_acc_max_f32), in place."""
        return self._c.spmm_acc_max(rowptr, col, val, B, C, E, rowmap, col_off, n_local, h_lo,
                                    plan=self._plan(rowptr, col, B.shape[0], B.shape[1]))

    def nonfinite_flag(self, x, flag):
        """flag |= 1 if x holds a NaN or an infinity (stream-ordered, no host sync)."""
        return self._c.nonfinite_flag(x, flag)

    def spmm_acc_min(self, rowptr, col, val, B, C, E, rowmap, col_off, precedes):
        """(C, E)[rowmap[r]] <- MIN step on the old pair and the min over row r, this product's columns first if
        ``precedes`` else last (dgs_spmm_csr_acc_min_f32), in place."""
        return self._c.spmm_acc_min(rowptr, col, val, B, C, E, rowmap, col_off, precedes,
                                    plan=self._plan(rowptr, col, B.shape[0], B.shape[1]))

    def spmm_acc_min_around(self, rowptr, col, val, B, C, E, rowmap, col_off, virt_lo, virt_n):
        """(C, E)[rowmap[r]] <- MIN over row r in row order, the old pair riding through the row as its virtual entry
        (dgs_spmm_csr_acc_min_around_f32), in place, one launch."""
        return self._c.spmm_acc_min_around(rowptr, col, val, B, C, E, rowmap, col_off, virt_lo, virt_n,
                                           plan=self._plan(rowptr, col, B.shape[0] + virt_n, B.shape[1]))

    def min_redo(self, rowmap, C, E, flag, rowptr, col, val, B):
        """Rows rowmap of (C, E) recomputed sequentially over the whole shard IF flag != 0 (stream-ordered; the redo-only
        form of dgs_spmm_min_merge_f32)."""
        return self._c.spmm_min_merge(rowmap, None, None, None, 0, None, C, E, flag, rowptr, col, val, B)

    def sddmm(self, rowptr, col, D1, D2, op=0, E=None):
        return self._c.sddmm(rowptr, col, D1, D2, op, E=E)

    def spmm_arg_backward(self, rowptr, col, val, E, grad, dense, need_dense=True, need_values=True):
        return self._c.spmm_arg_backward(rowptr, col, val, E, grad, dense, need_dense=need_dense, need_values=need_values)

    def gather_rows(self, src, ids):
        return self._c.gather_rows(src, ids)

    def scatter_add_rows(self, dst, ids, src):
        return self._c.scatter_add_rows(dst, ids, src)

    def relabel(self, ids, mapping):
        """ids -> mapping[ids] where ids >= 0, one pass (a torch chain of clamp/long/gather/where is five)."""
        return self._c.relabel_(ids.clone(), mapping)

    def csr2csc(self, rowptr, col, val, n_cols):
        """(colptr, row, values in CSC order | None, permutation CSC slot -> CSR slot)"""
        return self._c.csr2csc(rowptr, col, val, n_cols, want_perm=True)


_OPS = {'sum': 0, 'max': 1, 'min': 2, 'mean': 3}


class DistSpMM:
    """C_loc = reduce(A_loc_rows (*) B_global) with B row-partitioned like A.  ``ops`` is injectable so that the
    exchange logic can be exercised on CPU/gloo with a stand-in compute back end (tests only)."""

    def __init__(self, part: RowPartition, n_feat: int, ops=None, group=None, overlap: bool = True,
                 standalone: bool = False, min_form: Optional[str] = None):
        self.part, self.N, self.group, self.overlap = part, n_feat, group, overlap
        self.standalone = standalone
        self.ops = ops if ops is not None else _HipOps()
        # overlapped min after the exchange: 'around' = ONE accumulating launch (the local result a virtual entry of its row),
        # 'two' = the round-3 form (lower-rank slots folded in front of it, higher-rank ones behind: two launches)
        self.min_form = min_form or os.environ.get('DGS_DIST_MIN_FORM', 'around')
        if self.min_form not in ('around', 'two'):
            raise ValueError("min_form must be 'around' or 'two'")
        if not hasattr(self.ops, 'spmm_acc_min_around'):
            self.min_form = 'two'
        self.plan = HaloPlan(part, group, standalone)
        self.n_halo = self.plan.n_halo
        dev = part.col.device
        # one buffer for [local rows | halo rows]: the exchange lands directly where the kernel reads it
        self.B_ext = torch.empty((part.n_local + self.n_halo, n_feat), dtype=torch.float32, device=dev)
        t = torch.tensor([part.nnz], dtype=torch.int64, device=dev)
        if part.world > 1 and not standalone:
            if _staged(t, group):
                h = t.cpu()
                dist.all_reduce(h, group=group)
                t.copy_(h)
            else:
                dist.all_reduce(t, group=group)
        self.global_nnz = int(t.item())
        self._last_E = None      # global column ids of the last max/min: see the last_E property (mapped on first use)
        self.last_E_ext = None   # the same in the extended index space (what the backward needs)

    def imbalance(self) -> dict:
        """Per-rank load figures gathered over the group: nnz, rows, halo rows in, feature rows out (max / mean = the
        imbalance a step waits for).  Collective (one small all_gather); standalone engines report themselves only."""
        p, plan = self.part, self.plan
        mine = torch.tensor([p.nnz, p.n_local, self.n_halo, int(plan.send_ids.numel())], dtype=torch.int64,
                            device=p.col.device)
        if p.world > 1 and not self.standalone:
            src = mine.cpu() if _staged(mine, self.group) else mine
            allv = [torch.empty_like(src) for _ in range(p.world)]
            dist.all_gather(allv, src, group=self.group)
            t = torch.stack(allv).double()
        else:
            t = mine.double()[None]
        names = ('nnz', 'rows', 'halo_rows_in', 'rows_out')
        return {n: dict(max=int(t[:, i].max()), mean=float(t[:, i].mean()),
                        max_over_mean=round(float(t[:, i].max() / max(t[:, i].mean(), 1.0)), 3)) for i, n in enumerate(names)}

    def local_features(self) -> torch.Tensor:
        """View of the first n_local rows of the exchange buffer: fill it in place to skip the copy in spmm()."""
        return self.B_ext[:self.part.n_local]

    @property
    def last_E(self) -> Optional[torch.Tensor]:
        """Arg ids of the last max / min as GLOBAL column ids (-1 = none).  Mapped from the extended ids on first use: the
        backward works on the extended ids (``last_E_ext``), so a training step never pays for the pass over E
        (0.24 ms of a 0.86 ms min on a 2^20-row shard, N = 64)."""
        if self._last_E is None and self.last_E_ext is not None:
            self._last_E = self.ops.relabel(self.last_E_ext, self.plan.ext2glob32)
        return self._last_E

    def exchange(self, B_loc: torch.Tensor, async_op: bool = False):
        """Pack + all-to-all-v of the halo rows into B_ext[n_local:].  Returns (B_ext, work handle or None)."""
        p, plan = self.part, self.plan
        if B_loc.data_ptr() != self.B_ext.data_ptr():
            self.B_ext[:p.n_local].copy_(B_loc)
        work = None
        if p.world > 1 and not self.standalone:
            self._packed = self.ops.gather_rows(self.B_ext[:p.n_local], plan.send_ids)  # kept alive until waited
            work = _a2a(self.B_ext[p.n_local:], self._packed, plan.recv_splits, plan.send_splits, self.group, async_op)
        return self.B_ext, work

    def compute(self, reduce: str = 'sum', val: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The product over the CURRENT contents of the [local | halo] buffer (no exchange): one pass over the relabelled
        matrix.  ``val`` overrides the partition's edge values (same order as ``part.col``)."""
        val = self.part.val if val is None else val
        C, E = self.ops.spmm(_OPS[reduce], self.part.rowptr, self.plan.col_ext, val, self.B_ext)
        self._last_E = self.last_E_ext = None
        if E is not None:  # ext ids -> global column ids (-1 stays -1)
            self.last_E_ext = E
            self._last_E = None
        return C

    def spmm(self, B_loc: torch.Tensor, reduce: str = 'sum', val: Optional[torch.Tensor] = None) -> torch.Tensor:
        p, plan = self.part, self.plan
        if self.overlap and p.world > 1 and not self.standalone and reduce in ('sum', 'mean'):
            B_ext, work = self.exchange(B_loc, async_op=True)
            vl = plan.loc[2] if val is None else val[plan.nnz_pos_loc]
            vr = plan.rem[2] if val is None else val[plan.nnz_pos_rem]
            C, _ = self.ops.spmm(0, plan.loc[0], plan.loc[1], vl, B_ext[:p.n_local], shared_gpu=True)  # overlaps the exchange
            if work is not None:
                work.wait()  # current stream waits for the collective; the host does not block
            if plan.rem_rows.numel() > 0:
                # the halo product only covers the rows that have a remote entry and ACCUMULATES into them: no
                # temporary, no add pass (C[rem_rows[r]] += row r of A_rem . B_halo)
                self.ops.spmm_acc(plan.rem[0], plan.rem[1], vr, B_ext[p.n_local:], C, plan.rem_rows)
            if reduce == 'mean':
                C /= plan.deg[:, None]
            self._last_E = self.last_E_ext = None
            return C
        if self.overlap and p.world > 1 and not self.standalone and reduce == 'max' and plan.rows_sorted:
            # the same overlap for max: (value, arg) of the local columns while the halo travels, then the halo product
            # merges into them, ties going to the smaller GLOBAL column = the earlier entry of a sorted row (exact)
            B_ext, work = self.exchange(B_loc, async_op=True)
            vl = plan.loc[2] if val is None else val[plan.nnz_pos_loc]
            vr = plan.rem[2] if val is None else val[plan.nnz_pos_rem]
            C, E = self.ops.spmm(1, plan.loc[0], plan.loc[1], vl, B_ext[:p.n_local], shared_gpu=True)
            if work is not None:
                work.wait()
            if plan.rem_rows.numel() > 0:
                self.ops.spmm_acc_max(plan.rem[0], plan.rem[1], vr, B_ext[p.n_local:], C, E, plan.rem_rows, p.n_local,
                                      p.n_local, plan.h_lo)
            self.last_E_ext = E
            self._last_E = None
            return C
        if self.overlap and p.world > 1 and not self.standalone and reduce == 'min' and plan.rows_sorted:
            # min: the local (value, arg) while the halo travels; then ONE accumulating launch over the halo entries in which
            # the local result rides through each row as a virtual entry at the place of the local columns (min_form
            # 'around', round 5) - or, 'two', the lower-rank halo entries folded in FRONT of it and the higher-rank ones
            # BEHIND it by two launches (MIN keeps the later operand's bits on a tie, so only in-order folds are exact).  MIN cannot be folded across a NaN product at all, so features and
            # edge values are scanned for NaN / inf on the way (stream-ordered flag, no host sync) and the rows that have
            # remote entries are recomputed sequentially when the flag is up.  Cost of that guard: one scan of the local
            # features (under the exchange), one of the arrived halo, one of a caller-supplied ``val`` (the engine's own edge
            # values are scanned once and the verdict kept: they never change); a single infinity anywhere - a false alarm,
            # inf * 0 is the NaN the detector is after - sends every row with remote entries through the sequential redo,
            # which is correct but costs about what the one-pass min costs
            B_ext, work = self.exchange(B_loc, async_op=True)
            v_all = p.val if val is None else val.contiguous()  # (a strided override is fine for the products, not for the scan)
            vl = plan.loc[2] if val is None else val[plan.nnz_pos_loc]
            flag = torch.zeros(1, dtype=torch.int32, device=B_ext.device)
            C, E = self.ops.spmm(2, plan.loc[0], plan.loc[1], vl, B_ext[:p.n_local], shared_gpu=True)
            self.ops.nonfinite_flag(B_ext[:p.n_local], flag)
            if val is None and v_all is not None:
                if getattr(self, '_val_flag', None) is None:  # constant per engine: scanned once, OR-ed in from then on
                    self._val_flag = torch.zeros(1, dtype=torch.int32, device=B_ext.device)
                    self.ops.nonfinite_flag(v_all, self._val_flag)
                flag |= self._val_flag
            elif v_all is not None:
                self.ops.nonfinite_flag(v_all, flag)
            if work is not None:
                work.wait()
            if plan.rem_rows.numel() > 0:
                halo = B_ext[p.n_local:]
                self.ops.nonfinite_flag(halo, flag)
                vr = None if val is None else val[plan.nnz_pos_rem]
                if self.min_form == 'around' and p.n_local + self.n_halo < 2 ** 31 - 1:
                    sub, rows, pos = plan.min_around()
                    va = sub[2]
                    if vr is not None:
                        va = torch.where(pos >= 0, vr[pos.clamp(min=0)], torch.ones((), dtype=vr.dtype, device=vr.device))
                    if rows.numel() > 0:  # (like the two-launch form: an empty part is no launch)
                        self.ops.spmm_acc_min_around(sub[0], sub[1], va, halo, C, E, rows, p.n_local, plan.h_lo, p.n_local)
                else:
                    for (sub, rows, pos), first in zip(plan.min_parts(), (True, False)):
                        if rows.numel() > 0:
                            self.ops.spmm_acc_min(sub[0], sub[1], sub[2] if vr is None else vr[pos], halo, C, E, rows,
                                                  p.n_local, first)
                self.ops.min_redo(plan.rem_rows, C, E, flag, p.rowptr, plan.col_ext, v_all, B_ext)
            self.last_E_ext = E
            self._last_E = None
            return C
        self.exchange(B_loc)
        return self.compute(reduce, val)

    # ---- backward: the exchange reversed -----------------------------------------------------------------------------
    def _transposed(self, val=None):
        """(colptr, row, values in CSC order) of the relabelled local matrix; the structure is built once, caller-supplied
        values are permuted into it."""
        if getattr(self, '_csc', None) is None:
            p = self.part
            self._csc = self.ops.csr2csc(p.rowptr, self.plan.col_ext, p.val, p.n_local + self.n_halo)
        colptr, row, tval, perm = self._csc
        if val is not None:
            tval = val[perm.long()]
        return colptr, row, tval

    def _return_halo_grads(self, g_ext: torch.Tensor) -> torch.Tensor:
        """g_ext [n_local + n_halo, N] = gradient w.r.t. every row of the [local | halo] buffer.  The halo part belongs to
        other ranks: reverse all-to-all-v, then scatter-add into my rows (every send_ids row is unique per peer but one
        row may be wanted by several peers, so the adds are applied peer by peer - a fixed order, deterministic)."""
        p, plan = self.part, self.plan
        g_loc = g_ext[:p.n_local].contiguous()
        if p.world > 1 and not self.standalone:
            back = torch.empty((int(plan.send_ids.numel()), g_ext.shape[1]), dtype=torch.float32, device=g_ext.device)
            _a2a(back, g_ext[p.n_local:].contiguous(), plan.send_splits, plan.recv_splits, self.group)
            off = 0
            for n in plan.send_splits:  # one peer at a time: ids are unique inside a peer's block
                if n:
                    self.ops.scatter_add_rows(g_loc, plan.send_ids[off:off + n].contiguous(), back[off:off + n].contiguous())
                off += n
        return g_loc

    def _send_halo_grads(self, g_halo: torch.Tensor):
        """Starts the reversed all-to-all-v of the halo rows' gradients; returns (receive buffer, work handle)."""
        plan = self.plan
        back = torch.empty((int(plan.send_ids.numel()), g_halo.shape[1]), dtype=torch.float32, device=g_halo.device)
        self._g_halo = g_halo.contiguous()  # kept alive until waited
        work = _a2a(back, self._g_halo, plan.send_splits, plan.recv_splits, self.group, async_op=True)
        return back, work

    def _add_halo_grads(self, g_loc: torch.Tensor, back: torch.Tensor, work) -> torch.Tensor:
        if work is not None:
            work.wait()
        off = 0
        for n in self.plan.send_splits:  # one peer at a time, fixed order: deterministic
            if n:
                self.ops.scatter_add_rows(g_loc, self.plan.send_ids[off:off + n].contiguous(), back[off:off + n].contiguous())
            off += n
        return g_loc

    def spmm_sum_backward_dense(self, grad_C: torch.Tensor) -> torch.Tensor:
        """grad w.r.t. this rank's rows of B for the sum product: A_ext^T grad_C, halo parts sent home."""
        return self.backward(grad_C, 'sum', need_values=False)[0]

    def backward(self, grad_C: torch.Tensor, reduce: str = 'sum', need_dense: bool = True, need_values: bool = False,
                 val: Optional[torch.Tensor] = None, B_ext: Optional[torch.Tensor] = None,
                 E_ext: Optional[torch.Tensor] = None):
        """(grad of my rows of B or None, grad of my edge values or None) of ``spmm(B_loc, reduce)``; same semantics as
        the single-GPU operators (reference src/spmm.cpp:52-80,113-141,224-253 with the mean fix of SURVEY 3.4):
          sum   dB = A^T dC                 dW[e] = <dC[row e], B[col e]>
          mean  the same with dC rows scaled by 1/deg(row)
          max/min  only the arg entries carry gradient (E from the forward)
        B_ext / E_ext default to what the last forward left in the engine (the halo buffer is overwritten by the next
        forward: callers that interleave several forwards pass their own copies)."""
        p = self.part
        grad_C = grad_C.contiguous()
        val = p.val if val is None else val
        B_ext = self.B_ext if B_ext is None else B_ext
        g_loc = g_val = None
        if reduce in ('max', 'min'):
            E = self.last_E_ext if E_ext is None else E_ext
            assert E is not None, 'max/min backward needs the arg ids of the forward'
            gX, g_val = self.ops.spmm_arg_backward(p.rowptr, self.plan.col_ext, val, E, grad_C, B_ext,
                                                   need_dense=need_dense, need_values=need_values)
            if need_dense:
                g_loc = self._return_halo_grads(gX)
            return g_loc, g_val
        if reduce == 'mean':
            grad_C = grad_C / self.plan.deg[:, None]
        if need_values:
            g_val = self.ops.sddmm(p.rowptr, self.plan.col_ext, grad_C, B_ext)
        if need_dense:
            colptr, row, tval = self._transposed(None if val is p.val else val)
            if self.overlap and p.world > 1 and not self.standalone and self.n_halo > 0:
                # rows of A_ext^T = extended columns [local | halo slots], contiguous in the CSC arrays: the halo rows first,
                # their gradients start home (reversed all-to-all-v) while the local rows are computed, then the scatter-add
                nl = p.n_local
                if getattr(self, '_csc_cut', None) is None:
                    self._csc_cut = int(colptr[nl])  # once per engine
                    self._colptr_halo = (colptr[nl:] - self._csc_cut).contiguous()
                    self._colptr_loc = colptr[:nl + 1].contiguous()
                cut = self._csc_cut
                g_h, _ = self.ops.spmm(0, self._colptr_halo, row[cut:], None if tval is None else tval[cut:], grad_C)
                back, work = self._send_halo_grads(g_h)
                g_loc, _ = self.ops.spmm(0, self._colptr_loc, row[:cut], None if tval is None else tval[:cut], grad_C,
                                         shared_gpu=True)
                g_loc = self._add_halo_grads(g_loc.contiguous(), back, work)
            else:
                g_ext, _ = self.ops.spmm(0, colptr, row, tval, grad_C)  # [n_local + n_halo, N]
                g_loc = self._return_halo_grads(g_ext)
        return g_loc, g_val


class DistSpMMFn(torch.autograd.Function):
    """``C_loc = DistSpMMFn.apply(engine, B_loc, values, reduce)``: differentiable w.r.t. this rank's feature rows and
    (when ``values`` is a tensor that requires grad) its edge values.  ``values=None`` uses the partition's."""

    @staticmethod
    def forward(ctx, engine: 'DistSpMM', B_loc: torch.Tensor, values: Optional[torch.Tensor], reduce: str):
        ctx.engine, ctx.reduce = engine, reduce
        C = engine.spmm(B_loc, reduce, None if values is None else values.detach())
        need_v = values is not None and values.requires_grad
        # the halo buffer is reused by the next forward: keep what the values gradient / the arg backward reads
        keep_B = engine.B_ext.clone() if (need_v or reduce in ('max', 'min')) else None
        ctx.save_for_backward(values.detach() if values is not None else None, keep_B, engine.last_E_ext)
        return C

    @staticmethod
    def backward(ctx, grad_C):
        values, B_ext, E_ext = ctx.saved_tensors
        g_loc, g_val = ctx.engine.backward(grad_C, ctx.reduce, need_dense=ctx.needs_input_grad[1],
                                           need_values=ctx.needs_input_grad[2], val=values, B_ext=B_ext, E_ext=E_ext)
        return None, g_loc, g_val, None


class DistSpMMSum(torch.autograd.Function):
    """``C_loc = DistSpMMSum.apply(engine, B_loc)``: the sum product, differentiable w.r.t. B_loc (kept for callers of the
    first version of this module; DistSpMMFn is the general form)."""

    @staticmethod
    def forward(ctx, engine: 'DistSpMM', B_loc: torch.Tensor):
        ctx.engine = engine
        return engine.spmm(B_loc, 'sum')

    @staticmethod
    def backward(ctx, grad_C):
        return None, ctx.engine.spmm_sum_backward_dense(grad_C)
