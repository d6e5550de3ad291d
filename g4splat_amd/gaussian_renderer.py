"""`gaussian_renderer.render()` for MI355X -- the build's own counterpart of
2dgs/gaussian_renderer/__init__.py:19-166 (the reference Python never runs on the GPU box).

Same signature, same camera / model attributes read, same returned dict keys and shapes:
    render, viewspace_points, visibility_filter, radii                      (:110-114)
    rend_alpha, rend_normal, rend_normal_cam, rend_dist, surf_depth,
    surf_normal, surf_normal_cam, rend_depth                                (:155-164)
The rasterizer call goes to the HIP library through the drop-in `GaussianRasterizer`; the map
post-processing is plain torch for now.
"""
import math

import torch

from .diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer


def depths_to_points(view, depthmap):
    """2dgs/utils/point_utils.py:9-24: back-project a depth map to world-space points."""
    dev, dt = depthmap.device, depthmap.dtype
    c2w = (view.world_view_transform.T).inverse()
    W, H = int(view.image_width), int(view.image_height)
    ndc2pix = torch.tensor([[W / 2, 0, 0, W / 2], [0, H / 2, 0, H / 2], [0, 0, 0, 1]], dtype=dt, device=dev).T
    projection_matrix = c2w.T @ view.full_proj_transform
    intrins = (projection_matrix @ ndc2pix)[:3, :3].T
    gx, gy = torch.meshgrid(torch.arange(W, device=dev, dtype=dt), torch.arange(H, device=dev, dtype=dt), indexing="xy")
    pix = torch.stack([gx, gy, torch.ones_like(gx)], dim=-1).reshape(-1, 3)
    rays_d = pix @ intrins.inverse().T @ c2w[:3, :3].T
    rays_o = c2w[:3, 3]
    return depthmap.reshape(-1, 1) * rays_d + rays_o


def depth_to_normal(view, depth):
    """2dgs/utils/point_utils.py:26-37: normals from central differences of the back-projected depth."""
    points = depths_to_points(view, depth).reshape(*depth.shape[1:], 3)
    out = torch.zeros_like(points)
    dx = points[2:, 1:-1] - points[:-2, 1:-1]
    dy = points[1:-1, 2:] - points[1:-1, :-2]
    out[1:-1, 1:-1, :] = torch.nn.functional.normalize(torch.cross(dx, dy, dim=-1), dim=-1)
    return out


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, rasterizer_cls=None):
    """Render one view.  `rasterizer_cls` (default: the HIP GaussianRasterizer) exists only so that the
    CPU test-suite can drive this function's post-processing with the oracle; the product path never
    passes it."""
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
    rasterizer = (rasterizer_cls or GaussianRasterizer)(raster_settings=raster_settings)

    scales = rotations = cov3D_precomp = None
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
        scales, rotations = pc.get_scaling, pc.get_rotation

    shs = colors_precomp = None
    if override_color is None:
        shs = pc.get_features  # convert_SHs_python is forced off in the reference (:82)
    else:
        colors_precomp = override_color

    rendered_image, radii, allmap = rasterizer(means3D=xyz, means2D=screenspace_points, shs=shs,
                                               colors_precomp=colors_precomp, opacities=pc.get_opacity, scales=scales,
                                               rotations=rotations, cov3D_precomp=cov3D_precomp)
    rets = {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii}

    render_alpha = allmap[1:2]
    render_normal = allmap[2:5]
    render_normal_cam = render_normal.clone()
    render_normal = (render_normal.permute(1, 2, 0) @ (viewpoint_camera.world_view_transform[:3, :3].T)).permute(2, 0, 1)
    render_depth_median = torch.nan_to_num(allmap[5:6], 0, 0)
    render_depth_expected = torch.nan_to_num(allmap[0:1] / render_alpha, 0, 0)
    render_dist = allmap[6:7]
    surf_depth = render_depth_expected * (1 - pipe.depth_ratio) + pipe.depth_ratio * render_depth_median
    surf_normal = depth_to_normal(viewpoint_camera, surf_depth).permute(2, 0, 1)
    surf_normal = surf_normal * render_alpha.detach()  # rend_normal is un-normalised: weight alike
    surf_normal_cam = (surf_normal.clone().permute(1, 2, 0) @ viewpoint_camera.world_view_transform[:3, :3]).permute(2, 0, 1)
    rets.update({"rend_alpha": render_alpha, "rend_normal": render_normal, "rend_normal_cam": render_normal_cam,
                 "rend_dist": render_dist, "surf_depth": surf_depth, "surf_normal": surf_normal,
                 "surf_normal_cam": surf_normal_cam, "rend_depth": render_depth_expected})
    return rets
