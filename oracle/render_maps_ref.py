"""TEST INFRASTRUCTURE ONLY -- plain-torch restatement of the map post-processing of the reference's
`render()`; the checker for g4splat_amd/render_maps.py (HIP).  Never imported by the product.

PINNED: tests/golden/render_tail.npz holds the outputs AND the autograd gradient of the reference's own
`render()` (2d-gaussian-splatting/gaussian_renderer/__init__.py:19-166, which calls utils/point_utils.py:9-37), run on
the CPU of the build container with only the compiled rasterizer replaced by a stand-in that returns seeded tensors
(tests/golden/make_golden_maps.py); tests/test_oracle_golden.py checks this file against it, tests/
test_gpu_render_maps.py checks the HIP kernels against it.  This file follows the reference line by line:
    depths_to_points   2d-gaussian-splatting/utils/point_utils.py:9-24
    depth_to_normal    2d-gaussian-splatting/utils/point_utils.py:26-37
    render_maps        2d-gaussian-splatting/gaussian_renderer/__init__.py:117-164
"""
import torch


def depths_to_points(view, depthmap):
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
    points = depths_to_points(view, depth).reshape(*depth.shape[1:], 3)
    out = torch.zeros_like(points)
    dx = points[2:, 1:-1] - points[:-2, 1:-1]
    dy = points[1:-1, 2:] - points[1:-1, :-2]
    out[1:-1, 1:-1, :] = torch.nn.functional.normalize(torch.cross(dx, dy, dim=-1), dim=-1)
    return out


def render_maps(allmap, viewpoint_camera, depth_ratio):
    render_alpha = allmap[1:2]
    render_normal = allmap[2:5]
    render_normal_cam = render_normal.clone()
    render_normal = (render_normal.permute(1, 2, 0) @ (viewpoint_camera.world_view_transform[:3, :3].T)).permute(2, 0, 1)
    render_depth_median = torch.nan_to_num(allmap[5:6], 0, 0)
    render_depth_expected = torch.nan_to_num(allmap[0:1] / render_alpha, 0, 0)
    render_dist = allmap[6:7]
    surf_depth = render_depth_expected * (1 - depth_ratio) + depth_ratio * render_depth_median
    surf_normal = depth_to_normal(viewpoint_camera, surf_depth).permute(2, 0, 1)
    surf_normal = surf_normal * render_alpha.detach()
    surf_normal_cam = (surf_normal.clone().permute(1, 2, 0) @ viewpoint_camera.world_view_transform[:3, :3]).permute(2, 0, 1)
    return {"rend_alpha": render_alpha, "rend_normal": render_normal, "rend_normal_cam": render_normal_cam,
            "rend_dist": render_dist, "surf_depth": surf_depth, "surf_normal": surf_normal,
            "surf_normal_cam": surf_normal_cam, "rend_depth": render_depth_expected}
