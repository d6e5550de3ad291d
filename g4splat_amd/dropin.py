"""Registers this package's operators under the reference's module names so that unmodified G4Splat
code (`from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer`,
`from simple_knn._C import distCUDA2`; 2dgs/gaussian_renderer/__init__.py:14,
2dgs/scene/gaussian_model.py:21) picks up the MI355X implementation."""
import sys


def install():
    from . import diff_surfel_rasterization as dsr
    from . import simple_knn as sk
    from .simple_knn import _C as sk_c
    sys.modules.setdefault("diff_surfel_rasterization", dsr)
    sys.modules.setdefault("diff_surfel_rasterization._C", dsr._C)
    sys.modules.setdefault("simple_knn", sk)
    sys.modules.setdefault("simple_knn._C", sk_c)
    return dsr, sk
