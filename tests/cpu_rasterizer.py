"""TEST INFRASTRUCTURE: an oracle-backed stand-in with the GaussianRasterizer interface, on CPU tensors.
It lets the CPU suite exercise the host-side logic around the operator (render() post-processing, the
data-parallel gradient exchange) where no GPU exists.  Never imported by the product."""
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
