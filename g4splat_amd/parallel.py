"""One-view-per-GPU data parallelism for the render path (new functionality, the
reference trains one view per iteration on one GPU).

Every rank holds a replica of the six parameter tensors and renders its own view(s); ONE collective
per optimisation step sums the gradients: `torch.distributed` all-reduce (backend "nccl" = RCCL over
xGMI on ROCm, "gloo" in the CPU tests) over a single flat fp32 bucket that the parameters' `.grad`
tensors are views of -- 58 floats = 232 B per Gaussian, one large message instead of six small ones --
restricted to the rows of Gaussians visible on some rank when those are a minority (RowSparseAllReduce).  The densification side channel follows the reference's semantics: the per-view
||grad_means2D|| is taken BEFORE the reduction and then summed (gaussian_model.py:649-651 accumulates
a norm per view, not the norm of a sum), visibility counts are summed, radii are max-reduced
(train_with_refine_depth.py:583).
"""
import os
from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_views(views: Sequence, rank: int, world_size: int) -> List:
    """Round-robin assignment of a batch of training views: rank r renders views r, r+N, r+2N, ..."""
    return [v for i, v in enumerate(views) if i % world_size == rank]


class GradientBucket:
    """Flat gradient storage for a list of parameters; `.grad` of each parameter is a view into it."""

    def __init__(self, params: Iterable[torch.Tensor]):
        self.params = [p for p in params]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=torch.float32, device=ref.device)
        off = 0
        self.views = []
        for p in self.params:
            v = self.flat[off:off + p.numel()].view_as(p)
            p.grad = v  # autograd accumulates in place into the bucket
            self.views.append(v)
            off += p.numel()

    def zero(self):
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v  # optimizer.zero_grad(set_to_none=True) drops the views: re-attach


class RowSparseAllReduce:
    """SUM all-reduce of per-Gaussian rows that exchanges only the rows some rank can have touched.

    A view's gradients are non-zero only for the Gaussians visible in it (the rasterizer writes exact zeros for
    the others), typically 20-40 % of the scene.  After the MAX all-reduce of the radii (needed anyway for
    `max_radii2D`) every rank knows the same union-visibility mask; when the union is small the visible rows of
    all segments are packed into one persistent buffer, ONE all-reduce runs on that, and the rows are scattered
    back -- bit-identical to the dense all-reduce on those rows, and the remaining rows are zero on every rank.
    Over xGMI (per-link bound ring) the exchange is the scaling limiter, so bytes are what
    counts: at 2 ranks the union of two views is ~40 % of the rows.  Above `compact_below` the dense path runs.

    Only valid while rows outside the rank's visible set are zero (pure render gradients); pass
    `compact_below=0.0` to force the dense exchange when other loss terms write every row."""

    def __init__(self, flat: torch.Tensor, row_views: Sequence[torch.Tensor], group=None, compact_below: float = 0.7):
        self.flat, self.rows, self.group, self.compact_below = flat, list(row_views), group, compact_below
        self.P = self.rows[0].shape[0]
        self.widths = [int(r.shape[1]) for r in self.rows]
        self.compact = None  # persistent packed buffer, allocated on first use (capacity: every row)
        self.last_rows = None  # rows exchanged by the last reduce() (P for the dense path); for reporting
        self.last_dense = False  # the last reduce() all-reduced `flat` as a whole (row views outside it are NOT reduced)

    def reduce(self, union_mask: torch.Tensor):
        """`union_mask`: bool[P], identical on every rank (e.g. MAX-reduced radii > 0)."""
        idx = union_mask.nonzero(as_tuple=True)[0]  # one host read-back (the size); every rank gets the same rows
        n = int(idx.numel())
        if n > self.compact_below * self.P:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.last_rows, self.last_dense = self.P, True
            return
        self.last_rows, self.last_dense = n, False
        if n == 0:
            return
        if self.compact is None:
            self.compact = torch.empty(self.P * sum(self.widths), dtype=self.flat.dtype, device=self.flat.device)
        total = n * sum(self.widths)
        if self.flat.is_cuda:
            # one HIP kernel each way (g4s_pack_rows) instead of 2 x len(rows) index kernels
            self._pack(idx, n, unpack=0)
            dist.all_reduce(self.compact[:total], op=dist.ReduceOp.SUM, group=self.group)
            self._pack(idx, n, unpack=1)
            return
        off, parts = 0, []  # host tensors (gloo): plain torch indexing
        for r, w in zip(self.rows, self.widths):
            part = self.compact[off:off + n * w].view(n, w)
            torch.index_select(r, 0, idx, out=part)
            parts.append(part)
            off += n * w
        dist.all_reduce(self.compact[:off], op=dist.ReduceOp.SUM, group=self.group)
        for r, part in zip(self.rows, parts):
            r.index_copy_(0, idx, part)

    def _pack(self, idx, n, unpack):
        import ctypes
        from . import _lib
        lib = _lib.load()
        k = len(self.rows)
        for r in self.rows:
            if not r.is_contiguous():
                raise RuntimeError("RowSparseAllReduce: row views must be contiguous [P, k] tensors")
        ptrs = (ctypes.c_void_p * k)(*[r.data_ptr() for r in self.rows])
        widths = (ctypes.c_int * k)(*self.widths)
        dev = self.flat.device
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            rc = lib.g4s_pack_rows(k, ptrs, widths, ctypes.c_void_p(idx.data_ptr()), int(n),
                                   ctypes.c_void_p(self.compact.data_ptr()), int(unpack), stream)
        if rc != 0:
            raise RuntimeError(f"g4s_pack_rows failed ({rc}): {_lib.last_error()}")


class OwnerReduce:
    """Gradient exchange by OWNER-REDUCE over visible rows.

    The Gaussians are split into `world` contiguous index shards; rank d owns shard d.  One step = THREE collectives,
    all on persistent buffers (nothing is allocated per step; buffers only ever grow):

      begin(visible, radii)  right after the forward (the radii are known): the indices of this rank's visible rows
                       (fixed-size list, no host wait), their per-owner counts (two binary searches per owner), and
                       ONE all-reduce(MAX) over an int32 buffer [P + world^2] that carries the radii (-> max_radii2D,
                       train_with_refine_depth.py:583; bit 30 of a row = "visible on this rank", so the MAX also yields
                       the union of the visible sets) AND the count matrix (rank r fills row r, everything else is
                       zero, so MAX is a gather); the matrix tail goes to pinned host memory asynchronously.  All of
                       it overlaps the backward: the sizes are on the host long before they are needed.
      finish()         after the backward: ONE g4s_pack_rows launch packs this rank's visible rows row-major with their
                       index as a trailing int32 column, ONE all_to_all (uneven splits) sends them to the owners; the
                       owner adds what it received INTO ITS OWN SLICE of the gradient tensors, source by source in
                       ascending rank order (fixed order => bit-reproducible; one accumulating launch per source, the
                       indices read from the buffer); then (gather=True) the reduced shards are all-gathered IN
                       PLACE, one collective per tensor issued as one RCCL group, straight into the gradient tensors --
                       rank d's rows of a contiguous [P, w] tensor are contiguous.  (P not divisible by the world
                       size: ragged shards go through a padded staging buffer.)  When the rows visible on SOME rank are
                       fewer than `sparse_below` x P (known from begin()'s MAX-reduced radii, identically on every
                       rank), only those rows are gathered: owners pack them, ONE all_gather of equally padded
                       shards, every rank unpacks (`last_gather` says which form ran).  finish(gather=False) stops after
                       the owner's accumulation: ShardedAdam (below) then steps the owner's shard and gathers
                       PARAMETERS instead of gradients.

    Bytes through a rank's links per step, P Gaussians, w floats per row, visible fraction v, N ranks:
        all_to_all   v P (N-1)/N 4 (w + 1)      sent and received     (dense reduce-scatter: P (N-1)/N 4 w)
        all_gather   P (N-1)/N 4 w              received, P/N 4 w sent to each peer
        MAX          2 (N-1)/N 4 P              (ring all-reduce of the radii, 6 MB at P = 1.5 M; hidden behind the backward)
    At P = 1.5 M, w = 60, v = 0.28, N = 8: 90 MB + 315 MB instead of 2 x 315 MB for the dense all-reduce, and both
    are all-pairs patterns that use the seven xGMI links of a GPU concurrently (a ring all-reduce is bound by one).
    The result equals the dense all-reduce up to the order of the (at most N) additions per element; with two ranks it
    is bit-identical.  Rows that no rank sees stay exactly zero.

    Valid while a rank's gradient rows are zero outside its `visible` set (pure render gradients)."""

    def __init__(self, row_views: Sequence[torch.Tensor], group=None):
        self.rows, self.group = list(row_views), group
        for r in self.rows:
            if r.ndim != 2 or not r.is_contiguous() or r.dtype != torch.float32:
                raise RuntimeError("OwnerReduce: row views must be contiguous float32 [P, w] tensors")
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.P = int(self.rows[0].shape[0])
        self.widths = [int(r.shape[1]) for r in self.rows]
        self.width = sum(self.widths)
        self.shard = (self.P + self.world - 1) // self.world  # rows per owner (the last shard may be short)
        dev = self.rows[0].device
        self.dev = dev
        self.hip = dev.type == "cuda"
        self.even = self.P == self.world * self.shard  # equal shards: the reduced slices are gathered in place
        self.rccl = self.hip and dist.get_backend(group) == "nccl"
        W2 = self.world * self.world
        with (torch.cuda.device(dev) if self.hip else _nullctx()):
            # ---- persistent state (LAB_NOTES.md: collectives on persistent buffers only)
            self._meta = torch.zeros(self.P + W2, dtype=torch.int32, device=dev)  # radii | count matrix [src, dst]
            self._max_radii = torch.zeros(self.P, dtype=torch.int32, device=dev)  # what max_radii returns
            # sparse gather of the reduced shards (rows visible on SOME rank only): union mask / index list / per-owner counts
            self._union = torch.zeros(self.P, dtype=torch.bool, device=dev)
            self._uidx = torch.empty(self.P, dtype=torch.int64, device=dev)
            self._ucounts_host = (torch.zeros(self.world, dtype=torch.int32).pin_memory() if self.hip
                                  else torch.zeros(self.world, dtype=torch.int32))
            self._gin = self._gout = None  # float32 [capacity, width], grown on demand
            self._ragged_stage = {}  # (dtype, width) -> (stage, mine): all_gather_rows with ragged shards
            self._idx = torch.empty(self.P, dtype=torch.int64, device=dev)
            self._edges = torch.tensor([min(d * self.shard, self.P) for d in range(self.world + 1)], dtype=torch.int64,
                                       device=dev)
            self._counts_host = (torch.zeros(W2, dtype=torch.int32).pin_memory() if self.hip
                                 else torch.zeros(W2, dtype=torch.int32))
            self._event = torch.cuda.Event() if self.hip else None
            self._send = self._recv = None  # float32 [capacity, width + 1], grown on demand (never shrunk)
            self._gather = self._acc = None
            if not self.even:  # ragged shards: all_gather of padded shards through a staging buffer
                self._gather = torch.zeros(self.world * self.shard, self.width, device=dev)
                self._acc = torch.zeros(self.shard, self.width, device=dev)
        self._pending = False
        self._have_union = False
        # Gather only the rows some rank saw when they are a minority (sparse_below x P): on 2 / 4 ranks the union of the
        # step's views is 39 % / 70 % of S3's rows and ONE / three xGMI links carry a whole shard per step otherwise.
        self.sparse_below = 0.6
        # Rank order (round 6): a row's contributions are added in ascending RANK order whoever owns the row -- (((0 + g_0) +
        # g_1) + ...) + g_{N-1}, the owner's own rows in their place -- instead of "the owner's first, then the others".  That
        # is the order in which ONE process accumulating the same views one after the other sums them, so a step of N ranks
        # with one view each reproduces single-process gradient accumulation BIT FOR BIT (tests/test_gpu_dp.py, test_dp_gloo.py).
        # Costs nothing: the same one-launch kernel with its additions reordered (up to 8 sources = 9 ranks; beyond, and
        # with rank_order = False, the owner's rows come first).
        self.rank_order = True
        self.last_gather = None  # "dense" | "sparse": what the last finish(gather=True) did
        self._nonzero_static = hasattr(torch, "nonzero_static")
        # one RCCL group call for the per-tensor in-place gathers: probed ONCE, on a dummy tensor, and agreed on by all
        # ranks -- a rank that fell back on its own would issue a different collective sequence from its peers
        self._coalesce = self._probe_coalescing() if (self.rccl and not os.environ.get("G4S_OWNER_NO_COALESCE")) else False
        self.last_rows_sent = None
        self.allocations = 0  # buffer (re)allocations so far: stays constant once the exchange has warmed up
        self.timing = False   # True (HIP devices): bracket the pieces of begin() / finish() with events -> read_timers()
        self._timers = {}
        self.last_bytes = {}  # bytes this rank sent / received in the last step's collectives, by piece

    def _timed(self, name):
        """Context manager: with `timing` on, a pair of HIP events on the current stream around the piece `name` (the
        torch collectives are synchronous ops: the current stream waits for them, so the pair brackets the collective)."""
        return _EventPair(self, name) if (self.timing and self.hip) else _nullctx()

    def read_timers(self, reset=True):
        """-> {piece: mean ms per step} of everything recorded since the last reset (synchronises on the events)."""
        out = {}
        for name, pairs in self._timers.items():
            if pairs:
                pairs[-1][1].synchronize()
                out[name] = sum(a.elapsed_time(b) for a, b in pairs) / len(pairs)
        if reset:
            self._timers = {}
        return out

    # ---- helpers -------------------------------------------------------------------------------------------------
    def _probe_coalescing(self):
        """Can the per-tensor in-place all_gathers be issued as one RCCL group?  Decided in three steps so that no rank can
        leave its peers inside a half-issued group: (1) every rank checks LOCALLY, without any collective, that this torch
        has the coalescing manager and all_gather_into_tensor; (2) the flags are MIN-reduced -- the one collective every
        rank issues whatever its flag; (3) only if all ranks said yes is the grouped form run once on dummy tensors, and a
        failure there is fatal (it raises on the failing rank instead of silently desynchronising the sequence)."""
        ok = 1
        try:
            from torch.distributed.distributed_c10d import _coalescing_manager  # noqa: F401
            if not hasattr(dist, "all_gather_into_tensor"):
                ok = 0
        except ImportError:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=self.dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if not int(flag.item()):
            import warnings
            warnings.warn("OwnerReduce: coalesced in-place all_gather unavailable on some rank; one collective per tensor")
            return False
        from torch.distributed.distributed_c10d import _coalescing_manager
        a = torch.zeros(self.world, device=self.dev)
        b = torch.zeros(self.world, device=self.dev)
        with _coalescing_manager(group=self.group, device=self.dev, async_ops=False):
            dist.all_gather_into_tensor(a, a[self.rank:self.rank + 1], group=self.group)
            dist.all_gather_into_tensor(b, b[self.rank:self.rank + 1], group=self.group)
        return True

    def bounds(self, d=None):
        """[lo, hi): the rows owner `d` (default: this rank) holds."""
        d = self.rank if d is None else d
        return min(d * self.shard, self.P), min((d + 1) * self.shard, self.P)

    _bounds = bounds

    @property
    def max_radii(self):
        """int32 [P]: MAX over the ranks of the radii handed to the last begin().  A persistent buffer of its own, filled
        by finish() and valid until the NEXT finish() overwrites it (begin() does not touch it); clone it to keep it
        longer."""
        return self._max_radii

    def _buffer(self, which, rows):
        buf = getattr(self, which)
        if buf is None or buf.shape[0] < rows:
            cap = max(int(rows * 1.25) + 1024, 0 if buf is None else buf.shape[0])
            with (torch.cuda.device(self.dev) if self.hip else _nullctx()):
                buf = torch.empty(cap, self.width + 1, device=self.dev)
            setattr(self, which, buf)
            self.allocations += 1
        return buf

    def prepack(self, visible: torch.Tensor):
        """-> (rows, block_offs) for `_C.rasterize_gaussians_backward(..., out={"accumulate": "first", "packed": ...})`: the
        per-Gaussian kernel of the backward then writes this rank's visible rows straight into the exchange's send layout
        (row-major, index column last) and `finish(prepacked=True)` sends from there -- the pack launch between the backward
        and the all_to_all (0.08 ms at 1.5 M surfels) is gone.  Preconditions: ONE view per exchange on this rank (the rows
        hold that view's gradients), `visible` is exactly `radii > 0` of that view, and the row tensors are, in this order,
        dL_dmeans3D [P,3], dL_dsh [P,M,3] as [P,3M], dL_dopacity [P,1], dL_dscales [P,2], dL_drotations [P,4] and the
        view_stats [P,2] the same backward writes.  `rows` holds P rows (a persistent buffer; P x (3 M + 14) x 4 bytes)."""
        if self.widths[0] != 3 or self.widths[2:] != [1, 2, 4, 2] or self.widths[1] % 3:
            raise RuntimeError("OwnerReduce.prepack(): the row tensors must be means3D 3 | sh 3M | opacity 1 | scales 2 | "
                               f"rotations 4 | view_stats 2, got widths {self.widths}")
        if getattr(self, "_packed_all", None) is None:
            with (torch.cuda.device(self.dev) if self.hip else _nullctx()):
                self._packed_all = torch.empty(self.P, self.width + 1, device=self.dev)
                self._pad256 = torch.zeros((self.P + 255) // 256 * 256, dtype=torch.int32, device=self.dev)
                self._block_offs = torch.zeros((self.P + 255) // 256, dtype=torch.int32, device=self.dev)
            self.allocations += 1
        # visible Gaussians in the 256-blocks before each block (three small launches beside the backward)
        self._pad256[:self.P].copy_(visible)
        counts = self._pad256.view(-1, 256).sum(dim=1, dtype=torch.int32)
        torch.cumsum(counts, dim=0, out=self._block_offs)
        self._block_offs.sub_(counts)
        return self._packed_all, self._block_offs

    # ---- first half: sizes, overlapped with the backward -----------------------------------------------------------
    def begin(self, visible: torch.Tensor, radii: Optional[torch.Tensor] = None):
        """`visible`: bool[P], the rows this rank's views can have touched (radii > 0, OR-ed over its views);
        `radii`: optional integer [P] tensor whose MAX over the ranks is wanted (max_radii after finish())."""
        with (torch.cuda.device(self.dev) if self.hip else _nullctx()), self._timed("begin_local"):
            # fixed-size index list (padded with P, ascending => grouped by owner): the host does not wait for a count
            if self._nonzero_static:
                try:
                    torch.nonzero_static(visible, size=self.P, fill_value=self.P, out=self._idx.view(self.P, 1))
                except (TypeError, RuntimeError):
                    self._idx.copy_(torch.nonzero_static(visible, size=self.P, fill_value=self.P).view(-1))
            else:  # older torch: nonzero() synchronises on the count
                nz = visible.nonzero(as_tuple=True)[0]
                self._idx.fill_(self.P)
                self._idx[:nz.numel()] = nz
            # per-owner counts: the list is sorted, so owner d's rows are the range between two binary searches (a
            # scatter_add of 1.5 M ones onto 8 counters would serialise on them)
            pos = torch.searchsorted(self._idx, self._edges)
            if getattr(self, "debug_checks", False):
                self._visible_last = visible.clone()
            meta = self._meta
            # per row: the radius in the low 30 bits and "visible on this rank" in bit 30 -- after the MAX, bit 30 is the
            # union of the ranks' visible sets whatever the radii are, and the low bits are the largest radius among the
            # ranks that see the row (the others hold 0)
            if radii is not None:
                # (radii only where the rank sees the row: a row with radii > 0 but visible == False would otherwise lose
                # the MAX against another rank's 2^30 + smaller radius -- ADVICE r4)
                torch.mul(radii.to(torch.int32), visible.to(torch.int32), out=meta[:self.P])
                meta[:self.P].bitwise_or_(visible.to(torch.int32) << 30)
            else:
                torch.mul(visible.to(torch.int32), 1 << 30, out=meta[:self.P])
            tail = meta[self.P:].view(self.world, self.world)
            tail.zero_()
            tail[self.rank].copy_(pos[1:] - pos[:-1])
        with (torch.cuda.device(self.dev) if self.hip else _nullctx()):
            with self._timed("max_all_reduce"):
                dist.all_reduce(meta, op=dist.ReduceOp.MAX, group=self.group)  # radii MAX + the count matrix, one collective
            self.last_bytes["max_all_reduce"] = int(meta.numel() * 4 * 2 * (self.world - 1) / max(self.world, 1))
            # the union of the ranks' visible sets (every rank computes the same list from the same reduced radii) and
            # its per-owner counts, for the sparse gather of finish(); sizes reach the host with the count matrix
            self._have_union = self.world > 1
            if self._have_union:
                with self._timed("begin_union"):
                    torch.ge(meta[:self.P], 1 << 30, out=self._union)
                    if self._nonzero_static:
                        try:
                            torch.nonzero_static(self._union, size=self.P, fill_value=self.P, out=self._uidx.view(self.P, 1))
                        except (TypeError, RuntimeError):
                            self._uidx.copy_(torch.nonzero_static(self._union, size=self.P, fill_value=self.P).view(-1))
                    else:
                        nz = self._union.nonzero(as_tuple=True)[0]
                        self._uidx.fill_(self.P)
                        self._uidx[:nz.numel()] = nz
                    upos = torch.searchsorted(self._uidx, self._edges)
                    self._ucounts_dev = (upos[1:] - upos[:-1]).to(torch.int32)
                    self._ucounts_host.copy_(self._ucounts_dev, non_blocking=self.hip)
            if self.hip:
                self._counts_host.copy_(meta[self.P:], non_blocking=True)
                self._event.record(torch.cuda.current_stream(self.dev))
            else:
                self._counts_host.copy_(meta[self.P:])
        self._pending = True

    def _rows_kernel(self, idx, n, buf, mode):
        """g4s_pack_rows over all row views (include/g4s_rasterizer.h): mode 10 = pack row-major + index column,
        15 = unpack row-major, adding, indices from the buffer."""
        import ctypes
        from . import _lib
        lib = _lib.load()
        k = len(self.rows)
        ptrs = (ctypes.c_void_p * k)(*[r.data_ptr() for r in self.rows])
        widths = (ctypes.c_int * k)(*self.widths)
        with torch.cuda.device(self.dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
            rc = lib.g4s_pack_rows(k, ptrs, widths, ctypes.c_void_p(idx.data_ptr() if idx is not None else 0), int(n),
                                   ctypes.c_void_p(buf.data_ptr()), int(mode), stream)
        if rc != 0:
            raise RuntimeError(f"g4s_pack_rows failed ({rc}): {_lib.last_error()}")

    # sources from which the one-launch accumulation (a read + write of the whole shard) beats a launch per source
    # (read + write of the arrived rows only): measured at the metric size, tools/micro/owner_local_cost.py
    ONE_LAUNCH_SOURCES = 4

    def _accumulate_kernel(self, offs, cnts, buf, own_pos=0):
        """g4s_accumulate_rows[_ordered] (include/g4s_rasterizer.h): the rows of `buf` received from the sources (offs[i],
        cnts[i]) are added to my shard of the row views, source after source, in one launch; own_pos > 0: my own rows take
        that position in the order of additions (rank order)."""
        import ctypes
        from . import _lib
        lib = _lib.load()
        k, n = len(self.rows), len(offs)
        ptrs = (ctypes.c_void_p * k)(*[r.data_ptr() for r in self.rows])
        widths = (ctypes.c_int * k)(*self.widths)
        lo, hi = self.bounds()
        with torch.cuda.device(self.dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
            if own_pos:
                rc = lib.g4s_accumulate_rows_ordered(k, ptrs, widths, n, (ctypes.c_int * n)(*offs), (ctypes.c_int * n)(*cnts),
                                                     ctypes.c_void_p(buf.data_ptr()), int(lo), int(hi), int(own_pos), stream)
            else:
                rc = lib.g4s_accumulate_rows(k, ptrs, widths, n, (ctypes.c_int * n)(*offs), (ctypes.c_int * n)(*cnts),
                                             ctypes.c_void_p(buf.data_ptr()), int(lo), int(hi), stream)
        if rc != 0:
            raise RuntimeError(f"g4s_accumulate_rows failed ({rc}): {_lib.last_error()}")

    def all_gather_rows(self, tensors: Sequence[torch.Tensor]):
        """All-gathers the owners' row ranges of contiguous [P, ...] tensors in place (equal shards), as one RCCL group
        where the backend can: on return every rank holds every owner's rows."""
        lo, hi = self.bounds()
        if not self.even:
            for t in tensors:  # ragged: padded staging buffer per tensor
                w = t[0].numel() if t.ndim > 1 else 1
                flat = t.view(self.P, w)
                key = (t.dtype, w)
                if key not in self._ragged_stage:  # persistent per (dtype, width): nothing is allocated per step
                    self._ragged_stage[key] = (torch.zeros(self.world * self.shard, w, dtype=t.dtype, device=t.device),
                                               torch.zeros(self.shard, w, dtype=t.dtype, device=t.device))
                    self.allocations += 1
                stage, mine = self._ragged_stage[key]
                mine[:hi - lo] = flat[lo:hi]
                dist.all_gather_into_tensor(stage.view(-1), mine.view(-1), group=self.group)
                flat.copy_(stage[:self.P])
            return

        def issue():
            for t in tensors:
                mine = t[lo:hi].reshape(-1)  # a view: rank d's rows of a contiguous [P, w] tensor are contiguous
                # RCCL gathers in place when the input is the rank's own slot of the output; gloo wants them disjoint
                dist.all_gather_into_tensor(t.view(-1), mine if self.rccl else mine.clone(), group=self.group)
        if self._coalesce:
            from torch.distributed.distributed_c10d import _coalescing_manager
            with _coalescing_manager(group=self.group, device=self.dev, async_ops=False):
                issue()
        else:
            issue()

    # ---- second half: rows to their owners, owners accumulate, (optionally) everybody gets every shard ------------
    def finish(self, gather: bool = True, prepacked: bool = False, gather_rows: Optional[Sequence[int]] = None):
        """Reduces the rows in place.  gather=True: on return every row view holds the sum over all ranks.
        gather=False: only this rank's own rows [bounds()) do (what an owner-applied optimiser needs).
        prepacked=True: the backward has already written this rank's visible rows into prepack()'s buffer.
        gather_rows: indices of the row tensors the gather covers (default: all).  The others stay reduced on their
        OWNER only -- the densification statistics, which nobody reads between two densifications, are kept that way and
        gathered once per densification interval instead of once per step (ReplicatedDensification(defer_stats=True)).
        (When the sparse gather applies -- the union of the visible sets is small -- it moves every row tensor regardless.)

        PRECONDITION: a row this rank did not flag `visible` in begin() holds zeros here (what the rasterizer's backward
        leaves: every gradient of a Gaussian with radii == 0 is zero).  Only visible rows travel to their owners, and the
        sparse gather writes back only rows that SOME rank flagged -- a non-zero value in an unflagged row (a regulariser
        over all Gaussians added to .grad before the exchange, say) would neither be summed nor overwritten, and the
        replicas would drift apart silently.  Flag such rows visible, or set `debug_checks = True` to have every finish()
        verify the precondition (one reduction and a host read-back per call: for bring-up, not for the step)."""
        if not self._pending:
            raise RuntimeError("OwnerReduce.finish() without begin()")
        self._pending = False
        self.last_prepacked = bool(prepacked)
        if getattr(self, "debug_checks", False):
            hidden = ~self._visible_last
            bad = sum(int((r[hidden] != 0).any().item()) for r in self.rows)
            if bad:
                raise RuntimeError(f"OwnerReduce.finish(): {bad} of the row tensors hold non-zero values in rows that were "
                                   "not flagged visible in begin()")
        if self._event is not None:
            self._event.synchronize()  # recorded a whole backward ago: returns at once
        torch.bitwise_and(self._meta[:self.P], (1 << 30) - 1, out=self._max_radii)  # (the next begin() reuses _meta)
        mat = self._counts_host.view(self.world, self.world).tolist()
        send = [int(x) for x in mat[self.rank]]
        recv = [int(mat[s][self.rank]) for s in range(self.world)]
        # rows this rank owns itself stay where they are: they are neither packed nor sent
        a = sum(send[:self.rank])
        b = a + send[self.rank]
        total = sum(send)
        self.last_rows_sent = total - send[self.rank]
        send[self.rank] = 0
        recv[self.rank] = 0
        n, m = sum(send), sum(recv)
        W = self.width
        with (torch.cuda.device(self.dev) if self.hip else _nullctx()):
            if prepacked:
                # the backward's per-Gaussian kernel has written ALL of this rank's visible rows, in index order, into
                # prepack()'s buffer: the rows for owner d are the range [sum(counts[:d]), + counts[d]) of it.  My own rows
                # sit in the middle; all_to_all_single wants consecutive pieces, so they take part as a piece to myself (a
                # device-local copy of 1 / world of the rows inside the collective) and are skipped below -- my
                # contribution already sits in my slice of the row tensors.
                if getattr(self, "_packed_all", None) is None:
                    raise RuntimeError("OwnerReduce.finish(prepacked=True) without prepack()")
                if getattr(self, "debug_checks", False) and total:
                    # the rows were laid out by the backward from prepack()'s mask, the splits come from begin()'s: the
                    # index column the backward wrote has to be begin()'s index list, row for row
                    col = self._packed_all[:total, W].contiguous().view(torch.int32).to(torch.int64)
                    if not torch.equal(col, self._idx[:total]):
                        raise RuntimeError("OwnerReduce.finish(prepacked=True): the rows in prepack()'s buffer are not the rows "
                                           "begin() counted (different `visible` masks, or the backward did not run with "
                                           "out['packed'])")
                send[self.rank] = recv[self.rank] = b - a
                m = sum(recv)
                in_rows = self._buffer("_recv", m)[:m]
                with self._timed("all_to_all"):
                    dist.all_to_all_single(in_rows, self._packed_all[:total], recv, send, group=self.group)
            else:
                in_rows = self._buffer("_recv", m)[:m]
                out_rows = self._buffer("_send", n)[:n]
                # pack my visible rows, [n, width + 1] row-major: the rows for owner d are one contiguous range, and the
                # last column carries the row index (int32 bits), so rows and indices travel in ONE all_to_all
                if self.hip:
                    with self._timed("pack"):
                        if a:
                            self._rows_kernel(self._idx[:a], a, out_rows[:a], 10)
                        if total - b:
                            self._rows_kernel(self._idx[b:total], total - b, out_rows[a:], 10)
                else:
                    idx = torch.cat((self._idx[:a], self._idx[b:total]))
                    off = 0
                    for r, w in zip(self.rows, self.widths):
                        out_rows[:, off:off + w] = r.index_select(0, idx)
                        off += w
                    out_rows[:, W] = idx.to(torch.int32).view(torch.float32)
                with self._timed("all_to_all"):
                    dist.all_to_all_single(in_rows, out_rows, recv, send, group=self.group)
            self.last_bytes["all_to_all_sent"] = n * (W + 1) * 4          # (through the links: my own piece excluded)
            self.last_bytes["all_to_all_received"] = (m - recv[self.rank]) * (W + 1) * 4
            # owner: my own contribution already sits in my slice; add the other ranks' rows to it, source by source (a
            # source holds a row at most once => no duplicate indices inside one accumulation, and the order is fixed)
            o = 0
            with self._timed("accumulate"):
                if self.hip:
                    # one launch for all sources (g4s_accumulate_rows: the same additions in the same order as a launch
                    # per source, one read + write of my shard instead of one per source); every source's rows ascend by
                    # index -- the index list of begin() is sorted, and so are the rows the backward packs itself
                    offs, cnts, own_pos = [], [], 0
                    for s_ in range(self.world):
                        if recv[s_] and s_ != self.rank:  # (prepacked: my own rows came back to me; they are already in place)
                            offs.append(o)
                            cnts.append(recv[s_])
                            own_pos += 1 if s_ < self.rank else 0
                        o += recv[s_]
                    # rank order (see `rank_order`): my own rows join the sum behind the sources of lower rank.  With no such
                    # source (own first is rank order) or a single source (two addends commute) the plain forms already are.
                    ordered = self.rank_order and own_pos > 0 and 2 <= len(offs) <= 8
                    if ordered:
                        self._accumulate_kernel(offs, cnts, in_rows, own_pos)
                    elif len(offs) >= self.ONE_LAUNCH_SOURCES:
                        self._accumulate_kernel(offs, cnts, in_rows)
                    else:  # few sources: a launch each touches only the rows that arrived, not the whole shard
                        for o_, c_ in zip(offs, cnts):
                            self._rows_kernel(None, c_, in_rows[o_:o_ + c_], 15)
                else:
                    lo_, hi_ = self.bounds()
                    live = [s_ for s_ in range(self.world) if recv[s_] and s_ != self.rank]
                    below = [s_ for s_ in live if s_ < self.rank]
                    ordered = self.rank_order and below and len(live) >= 2
                    # host tensors: the same order of additions as the device path (rank order: the sources below me into a
                    # zero shard first, then my own rows, then the sources above me)
                    acc = [torch.zeros(hi_ - lo_, w, dtype=r.dtype) for r, w in zip(self.rows, self.widths)] if ordered else None
                    starts, o2 = {}, 0
                    for s_ in range(self.world):
                        starts[s_] = o2
                        o2 += recv[s_]

                    def add_source(s_, targets, shift):
                        c, o_ = recv[s_], starts[s_]
                        ridx = in_rows[o_:o_ + c, W].contiguous().view(torch.int32).to(torch.int64) - shift
                        off = 0
                        for r, w in zip(targets, self.widths):
                            r.index_add_(0, ridx, in_rows[o_:o_ + c, off:off + w])
                            off += w
                    if ordered:
                        for s_ in below:
                            add_source(s_, acc, lo_)
                        for a_, r in zip(acc, self.rows):
                            a_ += r[lo_:hi_]
                        for s_ in live:
                            if s_ > self.rank:
                                add_source(s_, acc, lo_)
                        for a_, r in zip(acc, self.rows):
                            r[lo_:hi_] = a_
                    else:
                        for s_ in live:
                            add_source(s_, self.rows, 0)
            if not gather:
                return
            # every rank gets every reduced shard
            lo, hi = self.bounds()
            if self._have_union:
                cu = [int(x) for x in self._ucounts_host.tolist()]
                if sum(cu) < self.sparse_below * self.P:
                    self._sparse_gather(cu)
                    return
            self.last_gather = "dense"
            if gather_rows is not None and len(set(gather_rows)) < len(self.rows):
                sub = [self.rows[i] for i in sorted(set(gather_rows))]
                Wg = sum(int(r.shape[1]) for r in sub)
                self.last_bytes["all_gather_sent_per_peer"] = (hi - lo) * Wg * 4
                self.last_bytes["all_gather_received"] = (self.P - (hi - lo)) * Wg * 4
                with self._timed("all_gather"):
                    self.all_gather_rows(sub)   # (ragged shards: per-tensor staging inside)
                return
            self.last_bytes["all_gather_sent_per_peer"] = (hi - lo) * W * 4
            self.last_bytes["all_gather_received"] = (self.P - (hi - lo)) * W * 4
            if self.even:
                with self._timed("all_gather"):
                    self.all_gather_rows(self.rows)
                return
            with self._timed("all_gather"):
                off = 0
                for r, w in zip(self.rows, self.widths):
                    self._acc[:hi - lo, off:off + w] = r[lo:hi]
                    off += w
                dist.all_gather_into_tensor(self._gather.view(-1), self._acc.view(-1), group=self.group)
                full = self._gather[:self.P]
                # (shards are padded to `shard` rows: rank d's rows sit at [d * shard, d * shard + (hi_d - lo_d)) = their global index)
                off = 0
                for r, w in zip(self.rows, self.widths):
                    r.copy_(full[:, off:off + w])
                    off += w


    def _sparse_gather(self, cu):
        """Gather of the reduced shards restricted to the rows some rank saw (`cu[d]` of them in owner d's shard, the
        same numbers and the same index list `_uidx` on every rank): owner d packs ITS union rows, one all_gather of
        shards padded to the largest count moves them, every rank writes the other owners' rows where they belong.
        Rows outside the union are zero on every rank and stay so.  Bit-identical to the dense gather."""
        W = self.width
        me = self.rank
        maxc = max(cu)
        offs = [0]
        for c in cu:
            offs.append(offs[-1] + c)
        self.last_gather = "sparse"
        self.last_bytes["all_gather_sent_per_peer"] = maxc * W * 4
        self.last_bytes["all_gather_received"] = (self.world - 1) * maxc * W * 4
        if maxc == 0:
            return

        def grow(name, rows):
            buf = getattr(self, name)
            if buf is None or buf.shape[0] < rows:
                buf = torch.empty(max(int(rows * 1.25) + 1024, 0 if buf is None else buf.shape[0]), W, device=self.dev)
                setattr(self, name, buf)
                self.allocations += 1
            return buf
        gin = grow("_gin", maxc)[:maxc]
        gout = grow("_gout", self.world * maxc)[:self.world * maxc]
        with self._timed("all_gather"):
            mine = self._uidx[offs[me]:offs[me] + cu[me]]
            if self.hip:
                if cu[me]:
                    self._rows_kernel(mine, cu[me], gin, 2)          # pack, row-major
            else:
                off = 0
                for r, w in zip(self.rows, self.widths):
                    gin[:cu[me], off:off + w] = r.index_select(0, mine)
                    off += w
            dist.all_gather_into_tensor(gout.view(-1), gin.view(-1), group=self.group)
            for s_ in range(self.world):
                if s_ == me or cu[s_] == 0:
                    continue
                idx = self._uidx[offs[s_]:offs[s_] + cu[s_]]
                part = gout[s_ * maxc:s_ * maxc + cu[s_]]
                if self.hip:
                    self._rows_kernel(idx, cu[s_], part, 3)          # unpack, row-major, overwrite
                else:
                    off = 0
                    for r, w in zip(self.rows, self.widths):
                        r.index_copy_(0, idx, part[:, off:off + w])
                        off += w


class _nullctx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class _EventPair:
    def __init__(self, owner, name):
        self.owner, self.name = owner, name

    def __enter__(self):
        self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.a.record(torch.cuda.current_stream(self.owner.dev))
        return self

    def __exit__(self, *exc):
        self.b.record(torch.cuda.current_stream(self.owner.dev))
        self.owner._timers.setdefault(self.name, []).append((self.a, self.b))
        return False


def adam_update_(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-15):
    """Plain Adam on host tensors, element by element what csrc/loss.hip's adam_kernel computes (torch.optim.Adam's
    single-tensor formula): used by ShardedAdam on CPU (gloo tests) -- on a HIP device the kernel itself runs."""
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    m.add_((g - m) * float(1.0 - beta1))
    v.mul_(float(beta2)).add_(g * g * float(1.0 - beta2))
    denom = v.sqrt() * float(1.0 / (bc2 ** 0.5)) + float(eps)
    p.sub_((m / denom) * float(lr / bc1))


class ShardedAdam:
    """Owner-applied Adam behind OwnerReduce (ZeRO-1).

    After `reducer.finish(gather=False)` rank d holds the fully reduced gradient of its own rows only.  It applies Adam
    to THOSE rows of the replicated parameters -- optimiser state (exp_avg, exp_avg_sq) exists for the rank's shard
    only: 2 x 232 B per Gaussian / N instead of per rank -- and the all_gather that would have carried 58 gradient
    floats per row carries the 58 updated parameter floats instead (same bytes, optimiser work / N).  Adam is
    element-wise, so the parameters equal the replicated optimiser's bit for bit (tests/test_dp_gloo.py).

        red = OwnerReduce(grad_row_views + [side]);  opt = ShardedAdam(params, grad_row_views, red, lrs, eps=1e-15)
        red.begin(visible, radii); ...backward...; red.finish(gather=False); opt.step(extra=[side])

    `params[i]` and `grads[i]` are contiguous float32 [P, ...] tensors; `extra` tensors ([P, ...], e.g. the
    densification statistics, reduced like the gradients) are gathered alongside the parameters.
    Densification changes P and with it the shard boundaries: `full_state()` returns the all-gathered state for the
    (replicated) row edits and `load_full_state()` re-shards it."""

    def __init__(self, params: Sequence[torch.Tensor], grads: Sequence[torch.Tensor], reducer: OwnerReduce,
                 lrs: Sequence[float], betas=(0.9, 0.999), eps: float = 1e-15):
        self.params, self.grads, self.red = list(params), list(grads), reducer
        self.lrs, self.betas, self.eps = [float(x) for x in lrs], (float(betas[0]), float(betas[1])), float(eps)
        if len(self.params) != len(self.grads) or len(self.params) != len(self.lrs) or not 1 <= len(self.params) <= 8:
            raise ValueError("ShardedAdam: 1..8 parameters, one gradient tensor and one learning rate each")
        lo, hi = reducer.bounds()
        for p, g in zip(self.params, self.grads):
            if p.shape != g.shape or p.shape[0] != reducer.P or not p.is_contiguous() or not g.is_contiguous() \
                    or p.dtype != torch.float32:
                raise ValueError("ShardedAdam: parameters and gradients must be contiguous float32 [P, ...] tensors")
        self.exp_avg = [torch.zeros_like(p[lo:hi]) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p[lo:hi]) for p in self.params]
        self.steps = 0

    @torch.no_grad()
    def step(self, lrs: Optional[Sequence[float]] = None, extra: Sequence[torch.Tensor] = ()):
        if lrs is not None:
            self.lrs = [float(x) for x in lrs]
        self.steps += 1
        lo, hi = self.red.bounds()
        ps = [p.data[lo:hi] for p in self.params]
        gs = [g[lo:hi] for g in self.grads]
        if hi > lo:
            if self.red.hip:
                import ctypes
                from . import _lib
                lib = _lib.load()
                k = len(ps)
                ptr = lambda ts: (ctypes.c_void_p * k)(*[t.data_ptr() for t in ts])
                with torch.cuda.device(self.red.dev):
                    stream = ctypes.c_void_p(torch.cuda.current_stream(self.red.dev).cuda_stream)
                    rc = lib.g4s_adam_step(k, ptr(ps), ptr(gs), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                                           (ctypes.c_longlong * k)(*[t.numel() for t in ps]),
                                           (ctypes.c_double * k)(*self.lrs), (ctypes.c_int * k)(*([self.steps] * k)),
                                           self.betas[0], self.betas[1], self.eps, stream)
                if rc != 0:
                    raise RuntimeError(f"g4s_adam_step failed ({rc}): {_lib.last_error()}")
            else:
                for p, g, m, v, lr in zip(ps, gs, self.exp_avg, self.exp_avg_sq, self.lrs):
                    adam_update_(p, g, m, v, lr, self.steps, self.betas[0], self.betas[1], self.eps)
        self.red.all_gather_rows([p.data for p in self.params] + list(extra))

    @torch.no_grad()
    def full_state(self):
        """-> (exp_avg, exp_avg_sq) lists of full [P, ...] tensors (all-gathered), for row edits at densification."""
        out = []
        lo, hi = self.red.bounds()
        for part in (self.exp_avg, self.exp_avg_sq):
            full = [torch.zeros_like(p.data) for p in self.params]
            for f, s_ in zip(full, part):
                f[lo:hi] = s_
            self.red.all_gather_rows(full)
            out.append(full)
        return out[0], out[1]

    @torch.no_grad()
    def load_full_state(self, params, grads, reducer, exp_avg, exp_avg_sq):
        """Re-shards after a densification: new parameter / gradient tensors, a reducer built for the new P, and the
        edited full state tensors."""
        self.params, self.grads, self.red = list(params), list(grads), reducer
        lo, hi = reducer.bounds()
        self.exp_avg = [t[lo:hi].clone() for t in exp_avg]
        self.exp_avg_sq = [t[lo:hi].clone() for t in exp_avg_sq]


class ViewParallel:
    """Gradient exchange of one optimisation step.

        vp = ViewParallel(model.parameters())            # after init_process_group
        for view in shard_views(batch, rank, world):     # local accumulation over this rank's views
            out = render(view, model, pipe, bg); loss(out).backward()
            vp.record_view(out["viewspace_points"], out["visibility_filter"], out["radii"])
        stats = vp.all_reduce()                           # ONE sum over xGMI + the small side channel
        optimizer.step(); vp.zero()

    A rank that holds several views of the batch (8 views over 1 / 2 / 4 GPUs) hands the loop to `accumulate()`, which keeps
    two of them in flight on HIP streams by default (pipeline.ViewPipeline; accumulation ordered, results bit-identical
    to the loop above):

        vp.accumulate(shard_views(batch, rank, world), view_step)   # view_step(view): render, loss, backward -> render()'s dict
    """

    def __init__(self, params: Iterable[torch.Tensor], group: Optional[dist.ProcessGroup] = None,
                 compact_below: float = 0.7, exchange: str = "allreduce"):
        """exchange = "allreduce" (one SUM all-reduce of the bucket, visible rows only when they are few) or "owner"
        (OwnerReduce: all_to_all of visible rows to index-shard owners + all_gather of the reduced shards)."""
        if exchange not in ("allreduce", "owner"):
            raise ValueError("exchange must be 'allreduce' or 'owner'")
        self.group = group
        self.exchange = exchange
        self._owner = None
        self._visible = None
        self.compact_below = compact_below  # RowSparseAllReduce threshold; 0.0 = always dense
        self._reducer = self._side = None
        self._pipe = self._pipe_key = None  # accumulate(): the ViewPipeline of the current (P, image size)
        self._capacity = {}   # (P, W, H) -> instance capacity of the pipeline's PresizedStates; 0 = unknown: stay sequential
        self._dirty = False   # something has been accumulated since zero()
        self.regrown = 0      # accumulate(): steps redone sequentially because a view outgrew the pipeline's capacity
        self.bucket = GradientBucket(params)
        n = self.bucket.params[0].shape[0]
        dev = self.bucket.flat.device
        self.grad_norm_sum = torch.zeros((n, 1), device=dev)
        self.vis_count = torch.zeros((n, 1), device=dev)
        self.max_radii = torch.zeros((n,), device=dev)

    @property
    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def record_view(self, viewspace_points, visibility_filter, radii):
        """Per-view densification statistics, taken before any reduction."""
        # (no boolean-mask indexing: `x[mask] += ...` reads the count back and would stall a pipelined loop.  The same
        # sums element by element: rows outside the filter get + 0.)
        self._dirty = True
        g = viewspace_points.grad
        vis = visibility_filter.unsqueeze(1)
        self.grad_norm_sum.add_(torch.where(vis, torch.norm(g, dim=-1, keepdim=True), torch.zeros_like(self.grad_norm_sum)))
        self.vis_count.add_(vis.to(self.vis_count.dtype))
        torch.maximum(self.max_radii, radii.to(self.max_radii.dtype), out=self.max_radii)
        if self._visible is None:
            self._visible = visibility_filter.clone()
        else:
            self._visible.logical_or_(visibility_filter)

    def accumulate(self, views, view_step, in_flight: int = 2, instance_capacity: Optional[int] = None,
                   check_overflow: bool = True, on_discard=None):
        """This rank's views of one optimisation step, accumulated locally (8 views over N < 8 GPUs).

            vp.accumulate(shard_views(batch, rank, world), view_step)     # then vp.all_reduce(); optimizer.step(); vp.zero()

        `view_step(view)` renders the view, forms its loss, calls `.backward()` and returns render()'s dict (this method
        records the view's densification statistics from it).  With more than one view on a HIP device the views go
        through a `pipeline.ViewPipeline` BY DEFAULT: `in_flight` of them on their own HIP streams (one view's
        launch-bound binning and HBM-bound per-Gaussian kernels beside another's VALU-bound blend kernels), the
        accumulation into `.grad` -- views of this object's bucket, persistent -- and the statistics ordered view after
        view, so that parameters after any number of steps equal the sequential loop's bit for bit
        (tests/test_gpu_dp.py).  `in_flight=1` is the plain sequential loop.

        The pipeline's PresizedStates need a bound on the (Gaussian, tile) instances of a view: `instance_capacity`,
        or -- by default -- 1.5 x the largest count seen in the first step at this (P, image size), which therefore runs
        sequentially through the reference-shaped forward (and every later step does if `view_step` does not go through
        this package's rasterizer front-end, so that no count is known).  Training draws other cameras every step, and
        a later view can bin more than that.  Such a view is invalid -- its forward kept only the first `capacity`
        instances -- and so is everything the step has added to `.grad` and to the statistics behind it: `check_overflow`
        (one read of the slots' status words per step, every view of the step included) then DISCARDS the step's local
        sums (zero()), grows the capacity to 1.5 x the count it has just seen, and redoes the step sequentially; the
        next step rebuilds the pipeline at the new size (`regrown` counts these steps).  That presumes the step's
        accumulation started in this call: if something was accumulated since zero() before it, there is nothing safe
        to discard and the overflow raises.

        CONTRACT for `view_step`: apart from the gradients it adds to this object's bucket it must be free of
        side effects that a second run would double -- the redo calls it again for EVERY view of the step.  What this method
        can undo itself it does: the bucket and the statistics are zeroed, and torch's random state (host generator and
        this device's) is put back to where the discarded pass started, so random backgrounds are drawn again as they
        were.  What it cannot see -- `.grad` of leaves outside the bucket (exposure / appearance modules), iteration
        counters, logged losses, hooks that fired on the truncated views' gradients -- is the caller's: `on_discard()` is
        called after the step's sums were dropped and before the redo, and the method returns {"regrown": True} for such a
        step ({"regrown": False} otherwise)."""
        views = list(views)
        params = self.bucket.params
        dev = params[0].device
        fresh = not self._dirty

        def plain():
            from . import diff_surfel_rasterization as dsr
            cap = 0
            for v in views:
                out = view_step(v)
                self.record_view(out["viewspace_points"], out["visibility_filter"], out["radii"])
                cap = max(cap, dsr.last_num_rendered() or 0)
            return cap

        sizes = {(int(v.image_width), int(v.image_height)) for v in views}
        if len(views) <= 1 or in_flight <= 1 or dev.type != "cuda" or len(sizes) != 1:
            plain()
            return {"regrown": False}
        from .pipeline import ViewPipeline
        (W, H), = sizes
        key = (int(params[0].shape[0]), W, H, min(int(in_flight), len(views)))
        if self._pipe_key != key:
            if self._pipe is not None:
                self._pipe.release_hooks()
                self._pipe = None
            cap = self._capacity.get(key[:3])
            if instance_capacity is not None:  # (a capacity that has had to grow stays grown)
                cap = max(int(instance_capacity), cap or 0)
            if cap is None:  # first step at this size: learn the instance counts, view after view
                seen = plain()
                # (0: view_step does not run this package's rasterizer front-end on this thread -- nothing to size by)
                self._capacity[key[:3]] = int(seen * 1.5) + 4096 if seen > 0 else 0
                return {"regrown": False}
            if cap == 0:
                plain()
                return {"regrown": False}
            self._pipe = ViewPipeline(key[0], W, H, int(cap), dev, k=key[3])
            self._pipe.order_accumulation(params)
            self._pipe_key = key
            try:  # the accumulation happens on the slots' streams on purpose (ordered by the hooks)
                torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
            except AttributeError:
                pass
        pipe = self._pipe
        rng = None
        if check_overflow:
            pipe.reset_peak()
            rng = (torch.get_rng_state(), torch.cuda.get_rng_state(dev))  # (host-side copies of the generator states: no device sync)
        for j, v in enumerate(views):
            with pipe.slot(j):
                out = view_step(v)
                pipe.after_previous_view()
                self.record_view(out["viewspace_points"], out["visibility_filter"], out["radii"])
        pipe.join()
        if check_overflow:
            seen, over = pipe.peak()
            if over:
                if not fresh:
                    raise RuntimeError("ViewParallel.accumulate: a view binned more instances than the pipeline's capacity "
                                       f"({pipe.states[0].capacity}) and the step's accumulation did not start in this call; "
                                       "pass a larger instance_capacity")
                # discard what the step has accumulated, grow, redo the step one view at a time
                self.zero()
                self._capacity[key[:3]] = int(seen * 1.5) + 4096
                pipe.release_hooks()
                self._pipe = self._pipe_key = None
                self.regrown += 1
                torch.set_rng_state(rng[0])
                torch.cuda.set_rng_state(rng[1], dev)
                if on_discard is not None:
                    on_discard()
                plain()
                return {"regrown": True}
        return {"regrown": False}

    def all_reduce(self, gather_stats: bool = True):
        """Sum the gradient bucket (and the statistics) over all ranks.  Returns the reduced statistics.
        gather_stats=False (owner exchange only): the gradients are gathered, the two statistics columns stay reduced on
        their owners (`side`, rows bounds()) -- see ReplicatedDensification(defer_stats=True); the returned grad_norm_sum /
        vis_count are then valid for this rank's own rows only."""
        if dist.is_initialized() and self.world_size > 1 and self.exchange == "owner":
            P = self.bucket.params[0].shape[0]
            self._ensure_owner()
            self._side[:, 0:1] = self.grad_norm_sum
            self._side[:, 1:2] = self.vis_count
            vis = self._visible if self._visible is not None else torch.zeros(P, dtype=torch.bool, device=self._side.device)
            # (a training loop that drives OwnerReduce itself calls begin() right after its forward, so that the size
            # exchange hides behind the backward -- bench.py does; here both halves run back to back)
            self._owner.begin(vis, radii=self.max_radii.to(torch.int32))  # the radii MAX rides in the same collective
            self._owner.finish(gather_rows=None if gather_stats else range(len(self.bucket.views)))
            self.max_radii = self._owner.max_radii.to(self.max_radii.dtype)
            self.grad_norm_sum, self.vis_count = self._side[:, 0:1].clone(), self._side[:, 1:2].clone()
        elif dist.is_initialized() and self.world_size > 1:
            dist.all_reduce(self.max_radii, op=dist.ReduceOp.MAX, group=self.group)
            if self._reducer is None:
                P = self.bucket.params[0].shape[0]
                self._side = torch.zeros((P, 2), device=self.bucket.flat.device)
                rows = [v.view(P, -1) for v in self.bucket.views]
                self._reducer = RowSparseAllReduce(self.bucket.flat, rows + [self._side], self.group, self.compact_below)
            self._side[:, 0:1] = self.grad_norm_sum
            self._side[:, 1:2] = self.vis_count
            self._reducer.reduce(self.max_radii > 0)
            if self._reducer.last_dense:  # the dense path reduced the bucket only
                dist.all_reduce(self._side, op=dist.ReduceOp.SUM, group=self.group)
            self.grad_norm_sum, self.vis_count = self._side[:, 0:1].clone(), self._side[:, 1:2].clone()
        return {"grad_norm_sum": self.grad_norm_sum, "vis_count": self.vis_count, "max_radii": self.max_radii}

    # ---- owner-applied optimiser (ZeRO-1) behind the same object ----------------------------------------------------
    def _ensure_owner(self):
        if self._owner is None:
            P = self.bucket.params[0].shape[0]
            self._side = torch.zeros((P, 2), device=self.bucket.flat.device)
            self._owner = OwnerReduce([v.view(P, -1) for v in self.bucket.views] + [self._side], self.group)
        return self._owner

    def sharded_adam(self, lrs, betas=(0.9, 0.999), eps=1e-15):
        """-> ShardedAdam over this object's parameters, gradient bucket and owner exchange (exchange="owner")."""
        if self.exchange != "owner":
            raise RuntimeError("ViewParallel.sharded_adam() needs exchange='owner'")
        return ShardedAdam([p.data for p in self.bucket.params], self.bucket.views, self._ensure_owner(), lrs, betas=betas, eps=eps)

    def reduce_to_owners(self):
        """First half of a ZeRO-1 step: the owner exchange up to the owners' accumulation (no gather).  Follow with
        `opt.step(extra=[vp.side])` -- the updated parameters and the reduced statistics are gathered together -- and
        `vp.stats_after_owner_step()`."""
        if self.exchange != "owner" or not dist.is_initialized():
            raise RuntimeError("ViewParallel.reduce_to_owners() needs exchange='owner' and an initialised process group")
        own = self._ensure_owner()
        P = self.bucket.params[0].shape[0]
        self._side[:, 0:1] = self.grad_norm_sum
        self._side[:, 1:2] = self.vis_count
        vis = self._visible if self._visible is not None else torch.zeros(P, dtype=torch.bool, device=self._side.device)
        own.begin(vis, radii=self.max_radii.to(torch.int32))
        own.finish(gather=False)

    @property
    def side(self):
        """[P, 2] (sum of the per-view gradient norms | visibility count): travels with the gradient rows."""
        self._ensure_owner()
        return self._side

    def stats_after_owner_step(self):
        self.max_radii = self._owner.max_radii.to(self.max_radii.dtype)
        self.grad_norm_sum, self.vis_count = self._side[:, 0:1].clone(), self._side[:, 1:2].clone()
        return {"grad_norm_sum": self.grad_norm_sum, "vis_count": self.vis_count, "max_radii": self.max_radii}

    def rebind(self, params):
        """After a densification the parameters are NEW tensors of another length: a new gradient bucket, statistics of the
        new length, and everything sized by P (the exchange's persistent buffers, the view pipeline) dropped -- the next
        step rebuilds them."""
        if self._pipe is not None:
            self._pipe.release_hooks()
        self._pipe = self._pipe_key = None
        self._owner = self._reducer = self._side = None
        self.bucket = GradientBucket(params)
        n, dev = self.bucket.params[0].shape[0], self.bucket.flat.device
        self.grad_norm_sum = torch.zeros((n, 1), device=dev)
        self.vis_count = torch.zeros((n, 1), device=dev)
        self.max_radii = torch.zeros((n,), device=dev)
        self._visible = None
        self._dirty = False

    def zero(self):
        self.bucket.zero()
        self.grad_norm_sum.zero_()
        self.vis_count.zero_()
        self.max_radii.zero_()
        self._visible = None
        self._dirty = False


def densify_generator(device, iteration: int, base_seed: int = 0) -> torch.Generator:
    """The generator every replica hands to densify_and_prune at `iteration` (gaussian_model.py:598: densify_and_split
    draws its children with torch.normal): seeded from (base_seed, iteration) alone, so all ranks -- and the single-process
    run with the same seed -- draw the same numbers without a broadcast."""
    g = torch.Generator(device=device)
    g.manual_seed((int(base_seed) * 1_000_003 + int(iteration)) & 0x7FFFFFFFFFFFFFFF)
    return g


class ReplicatedDensification:
    """`densify_and_prune` under data parallelism: every replica edits its rows itself and all of them end up with the same
    bits ("identical parameters after densify_and_split"; 2dgs/scene/gaussian_model.py:586-647 driven as in
    train_with_refine_depth.py:583-593).

    Everything the edit reads is replicated when it runs: the parameters (same bits after every exchange), the
    densification statistics (accumulated HERE from the REDUCED per-view sums, add_stats()), the Adam moments (replicated
    optimiser: as they are; ShardedAdam: all-gathered for the edit and re-sharded for the new P).  The one input that is
    not a function of those is densify_and_split's torch.normal: every rank draws from densify_generator(iteration) -- no
    communication (a broadcast of the edited rows would move 232 B per Gaussian over one root's links).  After the edit a
    digest of the new parameters is compared across the ranks (one MIN + one MAX all-reduce of three words) and a
    divergence raises instead of training on.

        dz = ReplicatedDensification(model, vp, sharded=opt)          # sharded=None: model.optimizer is the replicated Adam
        ... per step:  stats = vp.all_reduce()  |  vp.reduce_to_owners(); opt.step(extra=[vp.side]); stats = vp.stats_after_owner_step()
                       dz.add_stats(stats); vp.zero()
        ... every densification_interval:  dz.densify_and_prune(iteration, max_grad, min_opacity, extent, size_threshold)
    """

    def __init__(self, model, vp: "ViewParallel", sharded: Optional[ShardedAdam] = None, base_seed: int = 0, check: bool = True,
                 defer_stats: bool = False):
        """defer_stats=True (owner exchange): the two statistics columns are NOT gathered every step.  Every rank accumulates
        the reduced per-step sums of ITS OWN rows only (add_owner_stats() after the owners' accumulation; the step is
        `vp.reduce_to_owners(); opt.step()` or `vp.all_reduce(gather_stats=False)`), and densify_and_prune() starts by
        all-gathering the accumulated columns once -- 2 of the 60 floats per row leave the per-step gather (-3.3 % of its
        bytes), the same additions happen in the same order on the row's owner, so the statistics the edit reads are the same
        bits as with the per-step gather (tests/test_dp_gloo.py)."""
        self.model, self.vp, self.sharded, self.base_seed, self.check = model, vp, sharded, int(base_seed), check
        self.defer_stats = bool(defer_stats)
        self._stats_local = False   # defer_stats: the accumulators hold owner-local sums that have not been gathered yet

    @torch.no_grad()
    def add_stats(self, stats):
        """train_with_refine_depth.py:583-584 for the step's views at once: the per-view norms and counts were summed and
        the radii MAX-reduced by the exchange, identically on every rank."""
        m = self.model
        m.xyz_gradient_accum += stats["grad_norm_sum"]
        m.denom += stats["vis_count"]
        m.max_radii2D = torch.maximum(m.max_radii2D, stats["max_radii"].to(m.max_radii2D.dtype))

    @torch.no_grad()
    def add_owner_stats(self):
        """defer_stats: after the owners' accumulation of this step (vp.side holds the reduced sums in this rank's own
        rows), add them to this rank's rows of the model's accumulators; the radii MAX is replicated by the exchange's
        first collective as always."""
        m, own = self.model, self.vp._owner
        lo, hi = own.bounds()
        side = self.vp.side
        m.xyz_gradient_accum[lo:hi] += side[lo:hi, 0:1]
        m.denom[lo:hi] += side[lo:hi, 1:2]
        m.max_radii2D = torch.maximum(m.max_radii2D, own.max_radii.to(m.max_radii2D.dtype))
        self._stats_local = True

    @torch.no_grad()
    def gather_stats(self):
        """defer_stats: every rank holds the accumulated statistics of its own rows only; replicate them (one all_gather of
        two columns, once per densification interval).  densify_and_prune() calls this itself; a caller that derives its
        thresholds from the statistics calls it first.  Idempotent until the next add_owner_stats()."""
        if self.defer_stats and self._stats_local and dist.is_initialized() and self.vp.world_size > 1:
            self.vp._ensure_owner().all_gather_rows([self.model.xyz_gradient_accum, self.model.denom])
        self._stats_local = False

    def _digest(self):
        ps = self.model.parameters()
        dev = ps[0].device
        # (integer sums of the bit patterns: exact, order independent)
        words = [p.detach().contiguous().view(torch.int32).to(torch.int64).sum() for p in ps]
        return torch.stack([torch.tensor(ps[0].shape[0], dtype=torch.int64, device=dev), sum(words[:3]), sum(words[3:])])

    @torch.no_grad()
    def densify_and_prune(self, iteration, max_grad, min_opacity, extent, max_screen_size):
        m, opt = self.model, self.sharded
        self.gather_stats()
        if opt is not None:
            # moments live on their owners: gather them, let the row edits carry them as the model optimiser's state
            ea, es = opt.full_state()
            for p, a, b in zip(m.parameters(), ea, es):
                m.optimizer.state[p] = {"step": torch.tensor(float(opt.steps)), "exp_avg": a, "exp_avg_sq": b}
        m.densify_and_prune(max_grad, min_opacity, extent, max_screen_size,
                            generator=densify_generator(m._xyz.device, iteration, self.base_seed))
        params = m.parameters()
        self.vp.rebind(params)
        if opt is not None:
            st = [m.optimizer.state.pop(p) for p in params]
            opt.load_full_state([p.data for p in params], self.vp.bucket.views, self.vp._ensure_owner(),
                                [s_["exp_avg"] for s_ in st], [s_["exp_avg_sq"] for s_ in st])
        if self.check and dist.is_initialized() and self.vp.world_size > 1:
            d = self._digest()
            lo, hi = d.clone(), d.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.vp.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.vp.group)
            if not torch.equal(lo, hi):
                raise RuntimeError(f"ReplicatedDensification: the replicas differ after densify_and_prune at iteration "
                                   f"{iteration} (rows / digests min {lo.tolist()} max {hi.tolist()})")
        return params


def broadcast_parameters(params: Iterable[torch.Tensor], src: int = 0, group: Optional[dist.ProcessGroup] = None):
    """Make every replica identical (start of training, after a densification that draws random
    numbers: gaussian_model.py:598 `torch.normal` in densify_and_split)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        for p in params:
            dist.broadcast(p.data, src=src, group=group)
