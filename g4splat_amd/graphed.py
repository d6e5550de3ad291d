"""One training iteration as ONE HIP-graph launch (new functionality; the reference issues ~60 launches per iteration from
Python: train_with_refine_depth.py:373-420).

At the metric size the GPU is the limit and a graph changes nothing (LAB_NOTES.md).  On the reference's smaller scenes it is
the host: at 300 k surfels / 1200x680 an iteration takes 1.3 ms of which the GPU works 0.9 -- the Python between the
launches (render(), five autograd nodes, the optimiser) is the rest.  Everything on the path is capture-safe when the
rasterizer runs through a PresizedState (no host read-back) and the optimiser keeps its step counts and learning rates on
the device (FusedAdam(capturable=True)), so the whole iteration -- render, loss, backward, densification statistics, Adam
step, zero_grad -- is captured once per (P, resolution, intrinsics) and replayed:

    gaussians.training_setup(opt, capturable=True)
    step = TrainStepGraph(gaussians, body, example_camera, gt_shape=(3, H, W), instance_capacity=R_max)
    for it in range(...):
        gaussians.update_learning_rate(it)
        loss = step(camera, gt_image)              # copies the camera matrices + target into the graph's inputs, replays
        if it % densification_interval == 0:
            densify(...); step = TrainStepGraph(...)   # P changed: capture again (a few milliseconds)

`body(out, gt)` turns render()'s dict and the target into the scalar loss (any torch / g4splat_amd.losses code without
host synchronisation).  The graph's inputs are the camera's world_view_transform, full_proj_transform, camera_center and
the target image.  EVERYTHING ELSE is baked in at capture and needs a new TrainStepGraph when it changes: image size and
field of view (cameras with other intrinsics), the number of Gaussians and the storage of the optimiser's parameter
tensors (any densify / prune / reset_opacity that replaces them: a replay would train on freed memory), the active SH
degree (oneupSHdegree()), `pipe.depth_ratio` and whatever `body` closes over (loss weights).  The first three groups
are recorded at capture and checked on every call -- a stale graph raises instead of replaying.  `step.overflowed()` reads the PresizedState's overflow flag (synchronises): a frame that binned more
instances than `instance_capacity` is invalid and so is the update made from it.
"""
from types import SimpleNamespace

import torch

from . import _lib
from .diff_surfel_rasterization import _C, presized
from .gaussian_renderer import render


class TrainStepGraph:
    def __init__(self, gaussians, body, camera, gt_shape, instance_capacity, pipe=None, bg=None, warmup=3,
                 densification_stats=True):
        xyz = gaussians.get_xyz
        dev = xyz.device
        if not xyz.is_cuda:
            raise RuntimeError("TrainStepGraph: the model must live on a HIP device")
        opt = gaussians.optimizer
        if not all(g.get("capturable", False) for g in opt.param_groups):
            raise RuntimeError("TrainStepGraph: build the optimiser with capturable=True (training_setup(capturable=True))")
        self.gaussians, self.body, self.device = gaussians, body, dev
        self.pipe = pipe if pipe is not None else SimpleNamespace(depth_ratio=0.0, compute_cov3D_python=False)
        self.bg = bg if bg is not None else torch.zeros(3, device=dev)
        self.stats = densification_stats
        P, W, H = int(xyz.shape[0]), int(camera.image_width), int(camera.image_height)
        lib = _lib.load()
        # the graph's inputs
        self.cam = SimpleNamespace(image_width=W, image_height=H, FoVx=float(camera.FoVx), FoVy=float(camera.FoVy),
                                   world_view_transform=camera.world_view_transform.detach().clone(),
                                   full_proj_transform=camera.full_proj_transform.detach().clone(),
                                   camera_center=camera.camera_center.detach().clone(),
                                   znear=getattr(camera, "znear", 0.01), zfar=getattr(camera, "zfar", 100.0))
        self.gt = torch.zeros(tuple(gt_shape), dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self.state = _C.PresizedState(P, W, H, int(instance_capacity), dev)
            self.workspace = torch.empty(lib.g4s_rasterizer_backward_workspace(P, int(instance_capacity)), dtype=torch.uint8,
                                         device=dev)
            # warm-up iterations on the side stream (lazy optimiser state, allocator pools, autograd's own setup).  They are
            # real iterations on the example camera and a black target: everything they change is put back afterwards.
            keep = self._snapshot()
            for _ in range(max(int(warmup), 1)):
                self._iteration()
            self._restore(keep)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._iteration()
        self.replays = 0
        self._captured = self._signature()

    def _signature(self):
        """What a replay silently depends on besides its inputs: P, the active SH degree, depth_ratio and the addresses of
        the optimiser's parameter tensors."""
        g = self.gaussians
        params = [p for grp in g.optimizer.param_groups for p in grp["params"]]
        return (int(g.get_xyz.shape[0]), int(g.active_sh_degree), float(getattr(self.pipe, "depth_ratio", 0.0)),
                tuple(int(p.data_ptr()) for p in params))

    def _snapshot(self):
        g = self.gaussians
        params = [p for grp in g.optimizer.param_groups for p in grp["params"]]
        fresh = [p for p in params if len(g.optimizer.state[p]) == 0]  # their state is created by the warm-up: zero it after
        tensors = list(params)
        for p in params:
            tensors += [v for v in g.optimizer.state[p].values() if torch.is_tensor(v)]
        tensors += [t for t in (g.xyz_gradient_accum, g.denom, g.max_radii2D) if torch.is_tensor(t) and t.numel()]
        with torch.no_grad():
            return [(t, t.detach().clone()) for t in tensors], fresh

    def _restore(self, keep):
        saved, fresh = keep
        g = self.gaussians
        with torch.no_grad():
            for t, c in saved:
                t.copy_(c)
            for p in fresh:
                for v in g.optimizer.state[p].values():
                    if torch.is_tensor(v):
                        v.zero_()

    def _iteration(self):
        g = self.gaussians
        with presized(self.state, self.workspace):
            out = render(self.cam, g, self.pipe, self.bg)
            loss = self.body(out, self.gt)
            loss.backward()
        if self.stats:
            with torch.no_grad():
                g.add_densification_stats(out["viewspace_points"], out["visibility_filter"], out["radii"])
        g.optimizer.step()
        g.optimizer.zero_grad(set_to_none=True)
        return loss.detach()

    @torch.no_grad()
    def __call__(self, camera, gt):
        """One training iteration on `camera` (same image size and field of view as at capture) against `gt`.  Returns the
        loss tensor of the graph (overwritten by the next call)."""
        c = self.cam
        if (int(camera.image_width), int(camera.image_height)) != (c.image_width, c.image_height) or \
                abs(float(camera.FoVx) - c.FoVx) > 1e-12 or abs(float(camera.FoVy) - c.FoVy) > 1e-12:
            raise RuntimeError("TrainStepGraph was captured for another image size / field of view")
        now = self._signature()
        if now != self._captured:
            what = [n for n, a, b in zip(("number of Gaussians", "active SH degree", "pipe.depth_ratio",
                                          "optimiser parameter tensors (densify / prune / reset_opacity replaced them)"),
                                         self._captured, now) if a != b]
            raise RuntimeError("TrainStepGraph is stale -- changed since capture: " + ", ".join(what) +
                               "; build a new TrainStepGraph")
        c.world_view_transform.copy_(camera.world_view_transform, non_blocking=True)
        c.full_proj_transform.copy_(camera.full_proj_transform, non_blocking=True)
        c.camera_center.copy_(camera.camera_center, non_blocking=True)
        self.gt.copy_(gt, non_blocking=True)
        self.gaussians.optimizer.sync_lr()
        self.graph.replay()
        self.replays += 1
        return self.loss

    def overflowed(self):
        return int(self.state.status[3].item()) != 0

