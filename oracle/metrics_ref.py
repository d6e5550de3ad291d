"""TEST INFRASTRUCTURE (checker only; nothing under g4splat_amd/ imports it).  Torch restatements of the reference's
image metrics and photometric losses -- 2dgs/utils/loss_utils.py:17-79 (l1_loss, l1_loss_with_conf, l2_loss,
smooth_loss, the 11x11 Gaussian-window ssim), 2dgs/utils/image_utils.py:15-21 (mse, psnr) -- pinned to the reference's
own functions by tests/golden/losses.npz and losses_misc.npz.  The product's photometric loss is the fused HIP one
(g4splat_amd/losses.py -> csrc/loss.hip); this file is what it is checked against."""
import math

import torch
import torch.nn.functional as F


def l1_loss(network_output, gt):
    return torch.abs(network_output - gt).mean()


def l1_loss_with_conf(network_output, gt, conf):
    """loss_utils.py:20-24: confidence-weighted L1, normalised by the confidence mass."""
    return (torch.abs(network_output - gt) * conf).sum() / (conf.sum() + 1e-8)


def l2_loss(network_output, gt):
    return ((network_output - gt) ** 2).mean()


def smooth_loss(disp, img):
    """loss_utils.py:36-44: edge-aware second-order smoothness of a disparity map."""
    gdx = torch.abs(disp[:, 1:-1, :-2] + disp[:, 1:-1, 2:] - 2 * disp[:, 1:-1, 1:-1])
    gdy = torch.abs(disp[:, :-2, 1:-1] + disp[:, 2:, 1:-1] - 2 * disp[:, 1:-1, 1:-1])
    gix = torch.mean(torch.abs(img[:, 1:-1, :-2] - img[:, 1:-1, 2:]), 0, keepdim=True) * 0.5
    giy = torch.mean(torch.abs(img[:, :-2, 1:-1] - img[:, 2:, 1:-1]), 0, keepdim=True) * 0.5
    return (gdx * torch.exp(-gix)).mean() + (gdy * torch.exp(-giy)).mean()


def mse(img1, img2):
    return ((img1 - img2) ** 2).view(img1.shape[0], -1).mean(1, keepdim=True)


def psnr(img1, img2):
    mse = ((img1 - img2) ** 2).view(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


def _window(size, sigma, channel, like):
    g = torch.tensor([math.exp(-(x - size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(size)])
    g = (g / g.sum()).unsqueeze(1)
    w2 = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, size, size).contiguous().to(device=like.device, dtype=like.dtype)


def ssim(img1, img2, window_size=11, size_average=True):
    channel = img1.size(-3)
    w = _window(window_size, 1.5, channel, img1)
    pad = window_size // 2
    mu1 = F.conv2d(img1, w, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, w, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu12 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = F.conv2d(img1 * img1, w, padding=pad, groups=channel) - mu1_sq
    s2 = F.conv2d(img2 * img2, w, padding=pad, groups=channel) - mu2_sq
    s12 = F.conv2d(img1 * img2, w, padding=pad, groups=channel) - mu12
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return m.mean() if size_average else m.mean(1).mean(1).mean(1)
