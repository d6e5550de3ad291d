"""Fused map post-processing of `render()` (2dgs/gaussian_renderer/__init__.py:117-164 +
2dgs/utils/point_utils.py:9-37) over the C ABI of include/g4s_render_maps.h.

    maps = render_maps(allmap, viewpoint_camera, depth_ratio)
    -> dict(rend_alpha, rend_normal, rend_normal_cam, rend_depth, rend_dist, surf_depth, surf_normal, surf_normal_cam)

One HIP kernel forward, one backward (instead of ~15 element-wise torch kernels each way); torch is
plumbing (memory, stream, autograd graph).  No CPU path: host tensors raise like the rasterizer's.
"""
import ctypes

import torch

from . import _lib

_NAMES = ("rend_alpha", "rend_normal", "rend_normal_cam", "rend_depth", "rend_dist", "surf_depth", "surf_normal",
          "surf_normal_cam")
_CH = (1, 3, 3, 1, 1, 1, 3, 3)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def _mat(t, dev):
    return t.detach().to(device=dev, dtype=torch.float32).contiguous()


class _RenderMaps(torch.autograd.Function):
    @staticmethod
    def forward(ctx, allmap, world_view_transform, full_proj_transform, depth_ratio):
        # maps the loss does not use arrive as None in backward (-> NULL for the library) instead of as zero-filled
        # [.,H,W] tensors: five fills per training iteration less
        ctx.set_materialize_grads(False)
        if not allmap.is_cuda:
            raise RuntimeError("allmap must be a CUDA tensor")
        if allmap.ndim != 3 or allmap.size(0) != 7:
            raise RuntimeError("allmap must have dimensions (7, H, W)")
        lib = _lib.load()
        dev = allmap.device
        H, W = int(allmap.size(1)), int(allmap.size(2))
        am = allmap.detach().float().contiguous()
        wvt, fpt = _mat(world_view_transform, dev), _mat(full_proj_transform, dev)
        with torch.cuda.device(dev):
            outs = [torch.empty((c, H, W), dtype=torch.float32, device=dev) for c in _CH]
            nws = lib.g4s_render_maps_workspace()
            ws = torch.empty(nws, dtype=torch.uint8, device=dev)
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            rc = lib.g4s_render_maps_forward(W, H, _ptr(am), _ptr(wvt), _ptr(fpt), float(depth_ratio),
                                             *[_ptr(o) for o in outs], _ptr(ws), nws, stream)
        if rc != 0:
            raise RuntimeError(f"render_maps forward failed ({rc}): {_lib.last_error()}")
        ctx.save_for_backward(am, outs[5], wvt, fpt)
        ctx.depth_ratio = float(depth_ratio)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        am, surf_depth, wvt, fpt = ctx.saved_tensors
        lib = _lib.load()
        dev = am.device
        H, W = int(am.size(1)), int(am.size(2))
        gs = [None if g is None else g.float().contiguous() for g in grads]
        with torch.cuda.device(dev):
            g_allmap = torch.empty_like(am)
            nws = lib.g4s_render_maps_workspace()
            ws = torch.empty(nws, dtype=torch.uint8, device=dev)
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            rc = lib.g4s_render_maps_backward(W, H, _ptr(am), _ptr(surf_depth), _ptr(wvt), _ptr(fpt), ctx.depth_ratio,
                                              *[_ptr(g) for g in gs], _ptr(g_allmap), _ptr(ws), nws, stream)
        if rc != 0:
            raise RuntimeError(f"render_maps backward failed ({rc}): {_lib.last_error()}")
        return g_allmap, None, None, None


def render_maps(allmap, viewpoint_camera, depth_ratio):
    """The eight maps `render()` derives from the rasterizer's `allmap` (same names, shapes and autograd
    behaviour as the reference: alpha is detached inside surf_normal; camera matrices get no gradient)."""
    outs = _RenderMaps.apply(allmap, viewpoint_camera.world_view_transform, viewpoint_camera.full_proj_transform,
                             float(depth_ratio))
    return dict(zip(_NAMES, outs))
