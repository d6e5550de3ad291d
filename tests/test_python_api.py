"""Drop-in surface of the Python packages (dsr/diff_surfel_rasterization/__init__.py:158-222,
knn/ext.cpp): names, field order, argument validation and error behaviour -- no GPU needed."""
import inspect

import pytest
import torch

from g4splat_amd import dropin
from g4splat_amd.diff_surfel_rasterization import (GaussianRasterizationSettings, GaussianRasterizer,
                                                   rasterize_gaussians)


def _settings():
    return GaussianRasterizationSettings(image_height=32, image_width=48, tanfovx=0.5, tanfovy=0.4,
                                         bg=torch.zeros(3), scale_modifier=1.0, viewmatrix=torch.eye(4),
                                         projmatrix=torch.eye(4), sh_degree=0, campos=torch.zeros(3),
                                         prefiltered=False, debug=False)


def test_settings_fields_in_reference_order():
    assert GaussianRasterizationSettings._fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg",
                                                     "scale_modifier", "viewmatrix", "projmatrix", "sh_degree",
                                                     "campos", "prefiltered", "debug")


def test_forward_signature():
    sig = inspect.signature(GaussianRasterizer.forward)
    assert list(sig.parameters) == ["self", "means3D", "means2D", "opacities", "shs", "colors_precomp", "scales",
                                    "rotations", "cov3D_precomp"]
    assert list(inspect.signature(rasterize_gaussians).parameters) == [
        "means3D", "means2D", "sh", "colors_precomp", "opacities", "scales", "rotations", "cov3Ds_precomp",
        "raster_settings"]


def test_argument_validation_messages():
    r = GaussianRasterizer(_settings())
    x = torch.zeros(4, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=x, means2D=x, opacities=torch.ones(4, 1))
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=x, means2D=x, opacities=torch.ones(4, 1), shs=torch.zeros(4, 16, 3), colors_precomp=x)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=x, means2D=x, opacities=torch.ones(4, 1), colors_precomp=x)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=x, means2D=x, opacities=torch.ones(4, 1), colors_precomp=x, scales=torch.ones(4, 2),
          rotations=torch.ones(4, 4), cov3D_precomp=torch.ones(4, 9))


def test_cpu_tensors_are_rejected_like_the_reference():
    """CHECK_INPUT (rasterize_points.cu:27-28): no CPU fallback, ever."""
    r = GaussianRasterizer(_settings())
    x = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        r(means3D=x, means2D=x, opacities=torch.ones(4, 1), colors_precomp=x, scales=torch.ones(4, 2),
          rotations=torch.ones(4, 4))
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):
        r(means3D=torch.zeros(4, 2), means2D=x, opacities=torch.ones(4, 1), colors_precomp=x,
          scales=torch.ones(4, 2), rotations=torch.ones(4, 4))
    from g4splat_amd.simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        distCUDA2(torch.zeros(5, 3))


def test_dropin_module_names():
    """`import diff_surfel_rasterization` / `from simple_knn._C import distCUDA2` resolve to this package."""
    dropin.install()
    import diff_surfel_rasterization as d
    from simple_knn._C import distCUDA2
    from diff_surfel_rasterization import GaussianRasterizationSettings as S2
    assert d.GaussianRasterizer is GaussianRasterizer and S2 is GaussianRasterizationSettings
    assert callable(distCUDA2) and callable(d._C.rasterize_gaussians) and callable(d._C.mark_visible)
