"""One-view-per-GPU data parallelism for the render path (new functionality, the
reference trains one view per iteration on one GPU).

Every rank holds a replica of the six parameter tensors and renders its own view(s); ONE collective
per optimisation step sums the gradients: `torch.distributed` all-reduce (backend "nccl" = RCCL over
xGMI on ROCm, "gloo" in the CPU tests) over a single flat fp32 bucket that the parameters' `.grad`
tensors are views of -- 58 floats = 232 B per Gaussian, no packing copy, one large message instead of
six small ones.  The densification side channel follows the reference's semantics: the per-view
||grad_means2D|| is taken BEFORE the reduction and then summed (gaussian_model.py:649-651 accumulates
a norm per view, not the norm of a sum), visibility counts are summed, radii are max-reduced
(train_with_refine_depth.py:583).
"""
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


class ViewParallel:
    """Gradient exchange of one optimisation step.

        vp = ViewParallel(model.parameters())            # after init_process_group
        for view in shard_views(batch, rank, world):     # local accumulation over this rank's views
            out = render(view, model, pipe, bg); loss(out).backward()
            vp.record_view(out["viewspace_points"], out["visibility_filter"], out["radii"])
        stats = vp.all_reduce()                           # ONE sum over xGMI + the small side channel
        optimizer.step(); vp.zero()
    """

    def __init__(self, params: Iterable[torch.Tensor], group: Optional[dist.ProcessGroup] = None):
        self.group = group
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

    def all_reduce(self):
        """Sum the gradient bucket (and the statistics) over all ranks.  Returns the reduced statistics."""
        if dist.is_initialized() and self.world_size > 1:
            works = [dist.all_reduce(self.bucket.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
            side = torch.cat([self.grad_norm_sum, self.vis_count], dim=1)
            works.append(dist.all_reduce(side, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            works.append(dist.all_reduce(self.max_radii, op=dist.ReduceOp.MAX, group=self.group, async_op=True))
            for w in works:
                w.wait()
            self.grad_norm_sum, self.vis_count = side[:, 0:1].contiguous(), side[:, 1:2].contiguous()
        return {"grad_norm_sum": self.grad_norm_sum, "vis_count": self.vis_count, "max_radii": self.max_radii}

    def zero(self):
        self.bucket.zero()
        self.grad_norm_sum.zero_()
        self.vis_count.zero_()
        self.max_radii.zero_()


def broadcast_parameters(params: Iterable[torch.Tensor], src: int = 0, group: Optional[dist.ProcessGroup] = None):
    """Make every replica identical (start of training, after a densification that draws random
    numbers: gaussian_model.py:598 `torch.normal` in densify_and_split)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        for p in params:
            dist.broadcast(p.data, src=src, group=group)
