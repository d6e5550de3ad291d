"""Fused photometric loss of the training step (include/g4s_losses.h):

    loss, Ll1, ssim = photometric_loss(image, gt_image, lambda_dssim)
    # == (1 - l) * l1_loss(image, gt) + l * (1 - ssim(image, gt)),  l1_loss(...), ssim(...)
    #    2dgs/utils/loss_utils.py:17-18, 46-79;  train_with_refine_depth.py:382-383

One call computes the three scalars AND d loss / d image (two tiled HIP kernels + a fixed-order reduction);
the autograd backward only scales the stored gradient.  `Ll1` and `ssim` are returned for logging (detached,
as the reference uses them).  No CPU path."""
import ctypes

import torch

from . import _lib


class _Photometric(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt, lambda_dssim):
        if not image.is_cuda or not gt.is_cuda:
            raise RuntimeError("image and gt must be CUDA tensors")
        if image.ndim != 3 or image.size(0) != 3 or image.shape != gt.shape:
            raise RuntimeError("image and gt must both have dimensions (3, H, W)")
        lib = _lib.load()
        dev = image.device
        H, W = int(image.size(1)), int(image.size(2))
        x, y = image.detach().float().contiguous(), gt.detach().float().contiguous()
        need_grad = ctx.needs_input_grad[0]
        with torch.cuda.device(dev):
            out3 = torch.empty(3, dtype=torch.float32, device=dev)
            grad = torch.empty_like(x) if need_grad else None
            nws = lib.g4s_photometric_workspace(W, H)
            ws = torch.empty(nws, dtype=torch.uint8, device=dev)
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            rc = lib.g4s_photometric_loss(W, H, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()),
                                          float(lambda_dssim), ctypes.c_void_p(out3.data_ptr()),
                                          ctypes.c_void_p(grad.data_ptr() if need_grad else 0),
                                          ctypes.c_void_p(ws.data_ptr()), nws, stream)
        if rc != 0:
            raise RuntimeError(f"photometric_loss failed ({rc}): {_lib.last_error()}")
        ctx.grad = grad
        ctx.mark_non_differentiable(out3)
        return out3[0].clone(), out3

    @staticmethod
    def backward(ctx, g_loss, _g_out3):
        return (ctx.grad * g_loss if ctx.grad is not None else None), None, None


def photometric_loss(image, gt, lambda_dssim=0.2):
    """-> (loss, Ll1, ssim): `loss` carries the gradient to `image`; `Ll1`, `ssim` are detached scalars."""
    loss, out3 = _Photometric.apply(image, gt, float(lambda_dssim))
    return loss, out3[1], out3[2]


class _GeometryRegularizers(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rend_normal, surf_normal, rend_dist):
        for t in (rend_normal, surf_normal, rend_dist):
            if not t.is_cuda:
                raise RuntimeError("geometry_regularizers: CUDA tensors only")
        if (rend_normal.ndim != 3 or rend_normal.size(0) != 3 or rend_normal.shape != surf_normal.shape
                or rend_dist.numel() != rend_normal.size(1) * rend_normal.size(2)):
            raise RuntimeError("expected rend_normal, surf_normal (3, H, W) and rend_dist (1, H, W)")
        lib = _lib.load()
        dev = rend_normal.device
        H, W = int(rend_normal.size(1)), int(rend_normal.size(2))
        rn, sn = rend_normal.detach().float().contiguous(), surf_normal.detach().float().contiguous()
        rd = rend_dist.detach().float().contiguous()
        with torch.cuda.device(dev):
            out2 = torch.empty(2, dtype=torch.float32, device=dev)
            nws = lib.g4s_geometry_regularizers_workspace(W, H)
            ws = torch.empty(nws, dtype=torch.uint8, device=dev)
            rc = lib.g4s_geometry_regularizers_forward(
                W, H, ctypes.c_void_p(rn.data_ptr()), ctypes.c_void_p(sn.data_ptr()), ctypes.c_void_p(rd.data_ptr()),
                ctypes.c_void_p(out2.data_ptr()), ctypes.c_void_p(ws.data_ptr()), nws,
                ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"geometry_regularizers failed ({rc}): {_lib.last_error()}")
        ctx.save_for_backward(rn, sn)
        ctx.dist_shape = rend_dist.shape
        return out2

    @staticmethod
    def backward(ctx, g_out2):
        rn, sn = ctx.saved_tensors
        lib = _lib.load()
        dev = rn.device
        H, W = int(rn.size(1)), int(rn.size(2))
        g2 = g_out2.detach().float().contiguous()
        with torch.cuda.device(dev):
            d_rn, d_sn = torch.empty_like(rn), torch.empty_like(sn)
            d_rd = torch.empty(ctx.dist_shape, dtype=torch.float32, device=dev)
            rc = lib.g4s_geometry_regularizers_backward(
                W, H, ctypes.c_void_p(rn.data_ptr()), ctypes.c_void_p(sn.data_ptr()), ctypes.c_void_p(g2.data_ptr()),
                ctypes.c_void_p(d_rn.data_ptr()), ctypes.c_void_p(d_sn.data_ptr()), ctypes.c_void_p(d_rd.data_ptr()),
                ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"geometry_regularizers backward failed ({rc}): {_lib.last_error()}")
        return d_rn, d_sn, d_rd


def geometry_regularizers(rend_normal, surf_normal, rend_dist):
    """-> (normal_error.mean(), rend_dist.mean()) of train_with_refine_depth.py:391-396, i.e.
    (1 - (rend_normal * surf_normal).sum(dim=0)).mean() and rend_dist.mean(), as one fused HIP pass each way
    (include/g4s_losses.h).  The caller multiplies by lambda_normal / lambda_dist.  No CPU path."""
    out2 = _GeometryRegularizers.apply(rend_normal, surf_normal, rend_dist)
    return out2[0], out2[1]
