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

    The Gaussians are split into `world` contiguous index shards; rank d owns shard d.  One step:

      begin(visible)   right after the forward (the radii are known): the indices of this rank's visible rows, their
                       per-owner counts, an all_gather of the counts (world ints per rank) and an asynchronous copy of
                       the count matrix to the host -- all of it overlaps the backward, so the sizes are on the host
                       long before they are needed and nothing stalls on them (no `nonzero()` anywhere: the index list
                       has a fixed, padded size);
      finish()         after the backward: pack this rank's visible rows row-major (ONE g4s_pack_rows launch on a HIP
                       device), ONE all_to_all (uneven splits) of rows + ONE of their indices to the owners; the owner
                       adds what it received from the other ranks INTO ITS OWN SLICE of the gradient tensors, source by
                       source in ascending rank order (fixed order => bit-reproducible; one accumulating g4s_pack_rows
                       launch per source); then the reduced shards are all-gathered IN PLACE, one collective per
                       tensor straight into the gradient tensors (rank d's slice of a [P, w] tensor is contiguous) --
                       no staging buffer, no copy back.  (P not divisible by the world size: the shards are ragged and
                       the gather goes through a padded staging buffer instead.)

    Bytes through a rank's links per step, P Gaussians, w floats per row, visible fraction v, N ranks:
        all_to_all   v P (N-1)/N (4 w + 8)      sent and received     (dense reduce-scatter: P (N-1)/N 4 w)
        all_gather   P (N-1)/N 4 w              received, P/N 4 w sent to each peer
    At P = 1.5 M, w = 60, v = 0.28, N = 8: 91 MB + 315 MB instead of 2 x 315 MB for the dense all-reduce, and both
    collectives are all-pairs patterns that use the seven xGMI links of a GPU concurrently (a ring all-reduce is bound
    by one link).  Local HBM traffic on top of that: the packed rows once each way (2 x v P 4 w), nothing else.
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
        self._counts_host = None
        self._event = None
        self._idx = None
        self._gather = self._acc = None
        self._edges = torch.tensor([min(d * self.shard, self.P) for d in range(self.world + 1)], dtype=torch.int64, device=dev)
        if not self.even:  # ragged shards: all_gather of padded shards through a staging buffer
            self._gather = torch.zeros(self.world * self.shard, self.width, device=dev)
            self._acc = torch.zeros(self.shard, self.width, device=dev)
        self.rccl = self.hip and dist.get_backend(group) == "nccl"
        self._coalesce = self.rccl and not os.environ.get("G4S_OWNER_NO_COALESCE")
        self.last_rows_sent = None

    def _bounds(self, d):
        return min(d * self.shard, self.P), min((d + 1) * self.shard, self.P)

    def begin(self, visible: torch.Tensor):
        """`visible`: bool[P], the rows this rank's views can have touched (radii > 0, OR-ed over its views)."""
        # nonzero_static: fixed-size output (padded with P), so the host does not wait for the count here
        idx = torch.nonzero_static(visible, size=self.P, fill_value=self.P).view(-1)  # ascending => grouped by owner
        # per-owner counts: the list is sorted, so owner d's rows are the range between two binary searches (no atomics:
        # a scatter_add of 1.5 M ones onto 8 counters would serialise on them)
        pos = torch.searchsorted(idx, self._edges)
        counts = (pos[1:] - pos[:-1]).contiguous()
        mat = torch.zeros(self.world, self.world, dtype=torch.int64, device=self.dev)
        dist.all_gather_into_tensor(mat.view(-1), counts, group=self.group)  # mat[src, dst]
        self._idx = idx
        if self.hip:
            self._counts_host = torch.empty(mat.shape, dtype=mat.dtype, pin_memory=True)
            self._counts_host.copy_(mat, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()
        else:
            self._counts_host = mat.clone()

    def _rows_kernel(self, idx, n, buf, mode):
        """g4s_pack_rows over all row views: mode 2 = pack row-major, 7 = unpack row-major, adding (include/g4s_rasterizer.h)."""
        import ctypes
        from . import _lib
        lib = _lib.load()
        k = len(self.rows)
        ptrs = (ctypes.c_void_p * k)(*[r.data_ptr() for r in self.rows])
        widths = (ctypes.c_int * k)(*self.widths)
        with torch.cuda.device(self.dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
            rc = lib.g4s_pack_rows(k, ptrs, widths, ctypes.c_void_p(idx.data_ptr()), int(n),
                                   ctypes.c_void_p(buf.data_ptr()), int(mode), stream)
        if rc != 0:
            raise RuntimeError(f"g4s_pack_rows failed ({rc}): {_lib.last_error()}")

    def _gather_in_place(self, lo, hi):
        def issue():
            for r in self.rows:
                mine = r[lo:hi].reshape(-1)  # a view: rank d's rows of a contiguous [P, w] tensor are contiguous
                # RCCL gathers in place when the input is the rank's own slot of the output; gloo wants them disjoint
                dist.all_gather_into_tensor(r.view(-1), mine if self.rccl else mine.clone(), group=self.group)
        if self._coalesce:
            try:  # one RCCL group call for the per-tensor gathers
                from torch.distributed.distributed_c10d import _coalescing_manager
                with _coalescing_manager(group=self.group, device=self.dev, async_ops=False):
                    issue()
                return
            except (ImportError, RuntimeError, TypeError, NotImplementedError) as ex:
                self._coalesce = False  # this torch / backend cannot coalesce them: one collective per tensor
                import warnings
                warnings.warn(f"OwnerReduce: coalesced in-place all_gather unavailable ({ex}); issuing one collective per tensor")
        issue()

    def finish(self):
        """Reduces the rows in place: on return every row view holds the sum over all ranks."""
        if self._event is not None:
            self._event.synchronize()  # recorded a whole backward ago: returns at once
            self._event = None
        mat = self._counts_host.tolist()
        send = [int(x) for x in mat[self.rank]]
        recv = [int(mat[s][self.rank]) for s in range(self.world)]
        # rows this rank owns itself stay where they are: they are neither packed nor sent
        a = sum(send[:self.rank])
        b = a + send[self.rank]
        self.last_rows_sent = sum(send) - send[self.rank]
        idx = torch.cat((self._idx[:a], self._idx[b:sum(send)]))
        send[self.rank] = 0
        recv[self.rank] = 0
        n, m = sum(send), sum(recv)
        # pack my visible rows, [n, width] row-major: the rows for owner d are one contiguous range
        out_rows = torch.empty(n, self.width, device=self.dev)
        if self.hip:
            if n:
                self._rows_kernel(idx, n, out_rows, 2)
        else:
            off = 0
            for r, w in zip(self.rows, self.widths):
                out_rows[:, off:off + w] = r.index_select(0, idx)
                off += w
        in_rows = torch.empty(m, self.width, device=self.dev)
        in_idx = torch.empty(m, dtype=torch.int64, device=self.dev)
        dist.all_to_all_single(in_rows, out_rows, recv, send, group=self.group)
        dist.all_to_all_single(in_idx, idx, recv, send, group=self.group)
        # owner: my own contribution already sits in my slice; add the other ranks' rows to it, source by source (a
        # source holds a row at most once => no duplicate indices inside one accumulation, and the order is fixed)
        o = 0
        for s in range(self.world):
            c = recv[s]
            if c:
                if self.hip:
                    self._rows_kernel(in_idx[o:o + c], c, in_rows[o:o + c], 7)
                else:
                    off = 0
                    for r, w in zip(self.rows, self.widths):
                        r.index_add_(0, in_idx[o:o + c], in_rows[o:o + c, off:off + w])
                        off += w
            o += c
        # every rank gets every reduced shard
        lo, hi = self._bounds(self.rank)
        if self.even:
            self._gather_in_place(lo, hi)
            return
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


class ViewParallel:
    """Gradient exchange of one optimisation step.

        vp = ViewParallel(model.parameters())            # after init_process_group
        for view in shard_views(batch, rank, world):     # local accumulation over this rank's views
            out = render(view, model, pipe, bg); loss(out).backward()
            vp.record_view(out["viewspace_points"], out["visibility_filter"], out["radii"])
        stats = vp.all_reduce()                           # ONE sum over xGMI + the small side channel
        optimizer.step(); vp.zero()
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
        g = viewspace_points.grad
        self.grad_norm_sum[visibility_filter] += torch.norm(g[visibility_filter], dim=-1, keepdim=True)
        self.vis_count[visibility_filter] += 1
        self.max_radii = torch.maximum(self.max_radii, radii.to(self.max_radii.dtype))
        self._visible = visibility_filter.clone() if self._visible is None else (self._visible | visibility_filter)

    def all_reduce(self):
        """Sum the gradient bucket (and the statistics) over all ranks.  Returns the reduced statistics."""
        if dist.is_initialized() and self.world_size > 1 and self.exchange == "owner":
            P = self.bucket.params[0].shape[0]
            if self._owner is None:
                self._side = torch.zeros((P, 2), device=self.bucket.flat.device)
                self._owner = OwnerReduce([v.view(P, -1) for v in self.bucket.views] + [self._side], self.group)
            self._side[:, 0:1] = self.grad_norm_sum
            self._side[:, 1:2] = self.vis_count
            vis = self._visible if self._visible is not None else torch.zeros(P, dtype=torch.bool, device=self._side.device)
            self._owner.begin(vis)   # (a training loop calls begin() right after its forward; here both halves run back to back)
            self._owner.finish()
            dist.all_reduce(self.max_radii, op=dist.ReduceOp.MAX, group=self.group)
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

    def zero(self):
        self.bucket.zero()
        self.grad_norm_sum.zero_()
        self.vis_count.zero_()
        self.max_radii.zero_()
        self._visible = None


def broadcast_parameters(params: Iterable[torch.Tensor], src: int = 0, group: Optional[dist.ProcessGroup] = None):
    """Make every replica identical (start of training, after a densification that draws random
    numbers: gaussian_model.py:598 `torch.normal` in densify_and_split)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        for p in params:
            dist.broadcast(p.data, src=src, group=group)
