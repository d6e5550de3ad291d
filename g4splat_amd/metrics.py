"""The reference's image metrics under their own names, for the evaluation / logging lines of its training loop
(train_with_refine_depth.py:382, 715-716 use `l1_loss` and `psnr`; it imports `l1_loss_with_conf`, `ssim`):

    from g4splat_amd.metrics import l1_loss, l1_loss_with_conf, l2_loss, ssim, mse, psnr
    # utils/loss_utils.py:17-33, 46-79; utils/image_utils.py:15-21

`ssim` runs the fused HIP kernel (losses.photometric_loss with lambda = 1: loss = 1 - ssim, gradient included) and
therefore wants (3, H, W) HIP tensors; the others are one-line torch reductions that follow their inputs' device.
For the training loss itself use `losses.photometric_loss` -- one launch for L1, SSIM and the gradient."""
import torch

from .losses import photometric_loss


def l1_loss(network_output, gt):
    return (network_output - gt).abs().mean()


def l1_loss_with_conf(network_output, gt, conf):
    # weighted by the confidence map, normalised by its mass (loss_utils.py:20-24)
    return ((network_output - gt).abs() * conf).sum() / (conf.sum() + 1e-8)


def l2_loss(network_output, gt):
    return (network_output - gt).square().mean()


def mse(img1, img2):
    return (img1 - img2).square().reshape(img1.shape[0], -1).mean(1, keepdim=True)


def psnr(img1, img2):
    return 20 * torch.log10(1.0 / torch.sqrt(mse(img1, img2)))


def ssim(img1, img2):
    """11x11 Gaussian-window SSIM, mean over the image (loss_utils.py:46-79 with size_average=True); differentiable
    w.r.t. img1.  (3, H, W) or (1, 3, H, W) tensors on a HIP device."""
    if img1.ndim == 4 and img1.size(0) == 1:
        img1, img2 = img1[0], img2[0]
    loss, _l1, _ssim = photometric_loss(img1, img2, 1.0)
    return 1.0 - loss
