"""`gaussian_renderer.render()` for MI355X -- the build's own counterpart of
2dgs/gaussian_renderer/__init__.py:19-166 (the reference Python never runs on the GPU box).

Same signature, same camera / model attributes read, same returned dict keys and shapes:
    render, viewspace_points, visibility_filter, radii                      (:110-114)
    rend_alpha, rend_normal, rend_normal_cam, rend_dist, surf_depth,
    surf_normal, surf_normal_cam, rend_depth                                (:155-164)
`render_gslist` (:169-363) renders several models as one scene.
The rasterizer call goes to the HIP library through the drop-in `GaussianRasterizer`; the map
post-processing (:117-164 + utils/point_utils.py:9-37) is one fused HIP
kernel each way (`render_maps`).
"""
import math

import torch

from .diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from .render_maps import render_maps


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
    """Render one view (the reference's signature, :19).  The rasterizer and the map post-processing are the module
    attributes `GaussianRasterizer` and `render_maps` (the HIP implementations; there is no CPU path)."""
    xyz = pc.get_xyz
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=False)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    scales = rotations = cov3D_precomp = opacity = None
    if getattr(pipe, "compute_cov3D_python", False):
        # python formulation of T (:64-75); columns [0,1,3] only, so the z row of ndc2pix is irrelevant
        splat2world = pc.get_covariance(scaling_modifier)
        W, H = viewpoint_camera.image_width, viewpoint_camera.image_height
        near, far = viewpoint_camera.znear, viewpoint_camera.zfar
        ndc2pix = torch.tensor([[W / 2, 0, 0, (W - 1) / 2], [0, H / 2, 0, (H - 1) / 2], [0, 0, far - near, near],
                                [0, 0, 0, 1]], dtype=xyz.dtype, device=xyz.device).T
        world2pix = viewpoint_camera.full_proj_transform @ ndc2pix
        cov3D_precomp = (splat2world[:, [0, 1, 3]] @ world2pix[:, [0, 1, 3]]).permute(0, 2, 1).reshape(-1, 9)
    else:
        # (a model on the HIP device evaluates its three activations in one fused launch)
        act = getattr(pc, "get_activated", None) if xyz.is_cuda else None
        if act is not None:
            scales, rotations, opacity = act
        else:
            scales, rotations = pc.get_scaling, pc.get_rotation

    shs = colors_precomp = None
    if override_color is None:
        # convert_SHs_python is forced off in the reference (:82).  The reference concatenates the two SH parameter
        # tensors here (pc.get_features); the HIP rasterizer reads them where they are (split-SH entry points), which
        # saves the concatenation and the slicing of its gradient -- 4 x P x 192 B per iteration.
        split = getattr(pc, "get_features_split", None) if xyz.is_cuda else None
        shs = split if split is not None else pc.get_features
    else:
        colors_precomp = override_color

    rendered_image, radii, allmap = rasterizer(means3D=xyz, means2D=screenspace_points, shs=shs,
                                               colors_precomp=colors_precomp, opacities=pc.get_opacity if opacity is None else opacity, scales=scales,
                                               rotations=rotations, cov3D_precomp=cov3D_precomp)
    rets = {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii}

    rets.update(render_maps(allmap, viewpoint_camera, pipe.depth_ratio))
    return rets


class _JoinedModels:
    """The attributes `render()` reads, for a list of models rendered as one scene (render_gslist, :169-216: every
    activated tensor is concatenated in list order; the SH degree is the first model's)."""

    def __init__(self, pc_list):
        self._pcs = pc_list
        self.active_sh_degree = pc_list[0].active_sh_degree

    def _cat(self, name):
        return torch.cat([getattr(pc, name) for pc in self._pcs], dim=0)

    get_xyz = property(lambda self: self._cat("get_xyz"))
    get_opacity = property(lambda self: self._cat("get_opacity"))
    get_scaling = property(lambda self: self._cat("get_scaling"))
    get_rotation = property(lambda self: self._cat("get_rotation"))
    get_features = property(lambda self: self._cat("get_features"))

    def get_covariance(self, scaling_modifier=1):
        return torch.cat([pc.get_covariance(scaling_modifier) for pc in self._pcs], dim=0)


def render_gslist(viewpoint_camera, pc_list, pipe, bg_color, scaling_modifier=1.0, override_color=None):
    """2dgs/gaussian_renderer/__init__.py:169-363: the models of `pc_list` rendered together.  `override_color` may be
    a list with one tensor per model (:286-287).  Returns render()'s dictionary without the two `_cam` normal maps
    plus `model_start_indices` = [0, P_0, P_0 + P_1, ...] (:355-361)."""
    if not isinstance(pc_list, list):
        raise ValueError("pc_list must be a list of GaussianModel objects")
    if isinstance(override_color, list):
        override_color = torch.cat(override_color, dim=0)
    rets = render(viewpoint_camera, _JoinedModels(pc_list), pipe, bg_color, scaling_modifier, override_color)
    rets.pop("rend_normal_cam", None)
    rets.pop("surf_normal_cam", None)
    starts = [0]
    for pc in pc_list:
        starts.append(starts[-1] + int(pc.get_xyz.shape[0]))
    rets["model_start_indices"] = starts
    return rets
