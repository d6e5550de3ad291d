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
