"""TEST INFRASTRUCTURE: an oracle-backed stand-in with the GaussianRasterizer interface, on CPU tensors.
It lets the CPU suite exercise the host-side logic around the operator (render() post-processing, the
data-parallel gradient exchange) where no GPU exists.  Never imported by the product."""
import contextlib

import numpy as np
import torch

from oracle import oracle as om


class _OracleRasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs):
        n = lambda t: t.detach().cpu().numpy().astype(np.float32)
        o = om.Oracle()
        R, color, others, radii = o.rasterize_gaussians(
            n(rs.bg), n(means3D), n(colors_precomp), n(opacities), n(scales), n(rotations), rs.scale_modifier,
            n(cov3Ds_precomp), n(rs.viewmatrix), n(rs.projmatrix), rs.tanfovx, rs.tanfovy, rs.image_height,
            rs.image_width, n(sh), rs.sh_degree, n(rs.campos))
        ctx.o = o
        radii_t = torch.from_numpy(radii)
        ctx.mark_non_differentiable(radii_t)
        return torch.from_numpy(color), radii_t, torch.from_numpy(others)

    @staticmethod
    def backward(ctx, g_color, g_radii, g_depth):
        g = ctx.o.rasterize_gaussians_backward(g_color.numpy(), g_depth.numpy())
        t = torch.from_numpy
        return (t(g["means3D"]), t(g["means2D"]), t(g["sh"]), t(g["colors"]), t(g["opacity"]), t(g["scales"]),
                t(g["rotations"]), t(g["transMat"]), None)


class OracleRasterizer(torch.nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        e = torch.empty(0)
        return _OracleRasterize.apply(means3D, means2D, e if shs is None else shs,
                                      e if colors_precomp is None else colors_precomp, opacities,
                                      e if scales is None else scales, e if rotations is None else rotations,
                                      e if cov3D_precomp is None else cov3D_precomp, self.raster_settings)


@contextlib.contextmanager
def oracle_backend():
    """Inside the block `g4splat_amd.gaussian_renderer.render()` rasterizes with the CPU oracle and post-processes with
    the plain-torch restatement -- the CPU suite's way to drive render()'s host logic (and, on the GPU box, to
    produce the reference result the HIP render is compared with).  The product function has no hook for this: the
    two module attributes it calls are swapped from the outside."""
    from g4splat_amd import gaussian_renderer as gr
    from oracle.render_maps_ref import render_maps as maps_ref
    saved = gr.GaussianRasterizer, gr.render_maps
    gr.GaussianRasterizer, gr.render_maps = OracleRasterizer, maps_ref
    try:
        yield
    finally:
        gr.GaussianRasterizer, gr.render_maps = saved
