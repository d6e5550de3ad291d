"""Independent pure-PyTorch (autograd) surfel renderer -- BASELINE config 1's
"PyTorch-autograd CPU rasterizer".

TEST INFRASTRUCTURE ONLY (same rules as oracle/surfel_oracle.c).  It exists to pin the C
restatement from a second, differently structured implementation: it is vectorised over
pixels and loops over depth-sorted Gaussians (no tiles lists, no explicit backward), and
gradients come from autograd, so it checks
  * the forward outputs of the C oracle (colour, alpha, depth, normal, median depth,
    distortion; radii), and
  * every gradient of the explicit backward that is a true gradient
    (colour/SH, opacity when unclamped, normals, T-path to means/scales/rot).
Semantics follow dsr/cuda_rasterizer/forward.cu:75-253 (per Gaussian) and :346-419 (per
pixel), including the 16-px tile-rect clipping (auxiliary.h:66-76).
"""
import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]
NEAR, FAR = 0.2, 100.0


def sh_to_rgb(deg, shs, means3D, campos):
    """forward.cu:20-71 (same polynomial as 2dgs/utils/sh_utils.py:57-112), +0.5, clamp at 0."""
    d = means3D - campos[None]
    d = d / d.norm(dim=1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    r = SH_C0 * shs[:, 0]
    if deg > 0:
        r = r - SH_C1 * y * shs[:, 1] + SH_C1 * z * shs[:, 2] - SH_C1 * x * shs[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = (r + SH_C2[0] * xy * shs[:, 4] + SH_C2[1] * yz * shs[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * shs[:, 6]
             + SH_C2[3] * xz * shs[:, 7] + SH_C2[4] * (xx - yy) * shs[:, 8])
    if deg > 2:
        r = (r + SH_C3[0] * y * (3 * xx - yy) * shs[:, 9] + SH_C3[1] * xy * z * shs[:, 10]
             + SH_C3[2] * y * (4 * zz - xx - yy) * shs[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12]
             + SH_C3[4] * x * (4 * zz - xx - yy) * shs[:, 13] + SH_C3[5] * z * (xx - yy) * shs[:, 14]
             + SH_C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    return torch.clamp_min(r + 0.5, 0.0)


def quat_to_R(q):
    """auxiliary.h:212-234 on an (assumed unit) quaternion, *without* the normalisation, so that
    autograd reproduces quat_to_rotmat_vjp (auxiliary.h:237-281)."""
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    return R  # R[:, row, col]; columns are the splat axes


def render(means3D, scales, rotations, opacities, colors, viewmatrix, projmatrix, W, H, bg,
           scale_modifier=1.0, straight_through_clamp=True):
    """Returns (color[3,H,W], others[7,H,W], radii[P], aux) with autograd through every float input.

    `colors` is the per-Gaussian RGB ([P,3], e.g. from sh_to_rgb).  All tensors share one dtype.
    straight_through_clamp=True reproduces the reference's ungated gradient through
    min(0.99, .) (backward.cu:390)."""
    dt = means3D.dtype
    P = means3D.shape[0]
    R = quat_to_R(rotations)
    L0 = R[:, :, 0] * (scales[:, 0:1] * scale_modifier)
    L1 = R[:, :, 1] * (scales[:, 1:2] * scale_modifier)
    zeros = torch.zeros(P, 1, dtype=dt)
    M = torch.stack([torch.cat([L0, zeros], 1), torch.cat([L1, zeros], 1),
                     torch.cat([means3D, torch.ones(P, 1, dtype=dt)], 1)], dim=1)  # [P,3,4]
    C = M @ projmatrix  # clip coords of (u axis, v axis, centre)
    Tu = C[:, :, 0] * (W / 2) + C[:, :, 3] * ((W - 1) / 2)
    Tv = C[:, :, 1] * (H / 2) + C[:, :, 3] * ((H - 1) / 2)
    Tw = C[:, :, 3]
    p_view = torch.cat([means3D, torch.ones(P, 1, dtype=dt)], 1) @ viewmatrix
    normal = R[:, :, 2] @ viewmatrix[:3, :3]
    cosv = -(p_view[:, :3] * normal).sum(1)
    visible = (p_view[:, 2] > 0.2) & (cosv != 0)
    normal = torch.where((cosv > 0)[:, None], normal, -normal)

    # compute_aabb, forward.cu:119-147 (cutoff 3)
    t = torch.tensor([9.0, 9.0, -1.0], dtype=dt)
    dist = (Tw * Tw * t).sum(1)
    visible = visible & (dist != 0)
    f = t[None] / torch.where(dist == 0, torch.ones_like(dist), dist)[:, None]
    cx = (f * Tu * Tw).sum(1)
    cy = (f * Tv * Tw).sum(1)
    ex = torch.sqrt(torch.clamp_min(cx * cx - (f * Tu * Tu).sum(1), 1e-4))
    ey = torch.sqrt(torch.clamp_min(cy * cy - (f * Tv * Tv).sum(1), 1e-4))
    radius = torch.ceil(torch.maximum(ex, ey)).detach()
    gx, gy = (W + 15) // 16, (H + 15) // 16
    cxd, cyd = cx.detach(), cy.detach()

    def tile_lo(c, g):
        return torch.clamp(torch.trunc((c - radius) / 16), 0, g)

    def tile_hi(c, g):
        return torch.clamp(torch.trunc((c + radius + 15) / 16), 0, g)

    rx0, rx1, ry0, ry1 = tile_lo(cxd, gx), tile_hi(cxd, gx), tile_lo(cyd, gy), tile_hi(cyd, gy)
    visible = visible & ((rx1 - rx0) * (ry1 - ry0) > 0)
    radii = torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32)

    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    tyi, txi = torch.floor(ys / 16), torch.floor(xs / 16)

    T = torch.ones(H, W, dtype=dt)
    done = torch.zeros(H, W, dtype=torch.bool)
    Cc = torch.zeros(3, H, W, dtype=dt)
    Nn = torch.zeros(3, H, W, dtype=dt)
    Dd = torch.zeros(H, W, dtype=dt)
    M1 = torch.zeros(H, W, dtype=dt)
    M2 = torch.zeros(H, W, dtype=dt)
    distortion = torch.zeros(H, W, dtype=dt)
    median = torch.zeros(H, W, dtype=dt)
    mscale = FAR / (FAR - NEAR)

    depth_key = p_view[:, 2].detach().to(torch.float32)  # the reference sorts float32 bit patterns
    order = torch.sort(torch.where(visible, depth_key, torch.full_like(depth_key, float("inf"))), stable=True).indices
    for g in order.tolist():
        if not bool(visible[g]):
            continue
        in_rect = (txi >= rx0[g]) & (txi < rx1[g]) & (tyi >= ry0[g]) & (tyi < ry1[g])
        if not bool(in_rect.any()):
            continue
        k = xs[None] * Tw[g][:, None, None] - Tu[g][:, None, None]
        l = ys[None] * Tw[g][:, None, None] - Tv[g][:, None, None]
        p = torch.cross(k, l, dim=0)
        ok = in_rect & (~done) & (p[2] != 0)
        pz = torch.where(p[2] == 0, torch.ones_like(p[2]), p[2])
        sx, sy = p[0] / pz, p[1] / pz
        rho3d = sx * sx + sy * sy
        ddx, ddy = cx[g] - xs, cy[g] - ys
        rho2d = 2.0 * (ddx * ddx + ddy * ddy)
        use3d = rho3d <= rho2d
        rho = torch.where(use3d, rho3d, rho2d)
        depth = torch.where(use3d, sx * Tw[g, 0] + sy * Tw[g, 1] + Tw[g, 2], Tw[g, 2].expand_as(sx))
        ok = ok & (depth >= NEAR)
        G = torch.exp(-0.5 * rho)
        a_raw = opacities[g, 0] * G
        if straight_through_clamp:
            alpha = a_raw + (torch.clamp(a_raw, max=0.99) - a_raw).detach()
        else:
            alpha = torch.clamp(a_raw, max=0.99)
        ok = ok & (alpha >= 1.0 / 255.0)
        test_T = T * (1 - alpha)
        newly_done = ok & (test_T < 0.0001)
        done = done | newly_done
        ok = ok & ~newly_done
        w = torch.where(ok, alpha * T, torch.zeros_like(T))
        safe_depth = torch.where(ok, depth, torch.ones_like(depth))
        m = mscale * (1 - NEAR / safe_depth)
        A = 1 - T
        distortion = distortion + (m * m * A + M2 - 2 * m * M1) * w
        Dd = Dd + safe_depth * w
        M1 = M1 + m * w
        M2 = M2 + m * m * w
        median = torch.where(ok & (T > 0.5), depth, median)
        Nn = Nn + normal[g][:, None, None] * w[None]
        Cc = Cc + colors[g][:, None, None] * w[None]
        T = torch.where(ok, test_T, T)
    color = Cc + T[None] * bg[:, None, None]
    others = torch.cat([Dd[None], (1 - T)[None], Nn, median[None], distortion[None]], 0)
    aux = dict(Tu=Tu, Tv=Tv, Tw=Tw, center=torch.stack([cx, cy], 1), normal=normal, visible=visible)
    return color, others, radii, aux
