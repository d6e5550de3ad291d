"""Makes the reference's Python importable on this CPU-only build container (golden generation only; nothing here
ever runs on the GPU box or inside a test).

The reference modules that restate the hot path's neighbours import packages this image lacks (`cv2`,
`matplotlib`, `plyfile`, the compiled `simple_knn._C` and `diff_surfel_rasterization`) without using them on the
paths we record, and they hard-code `device="cuda"` / `.cuda()`.  `reference_modules()`:
  * registers empty stand-in modules for the absent imports (attributes the generators need -- the rasterizer
    class -- are injected by the caller),
  * registers `scene` as a bare namespace package so that `scene.gaussian_model` loads without executing
    scene/__init__.py (which pulls in the dataset readers),
  * maps every explicit `device="cuda"` of the torch factory functions to the CPU and turns `Tensor.cuda()` into
    the identity.
Nothing of the reference is copied: its own files are imported from /root/reference and called."""
import contextlib
import sys
import types

import torch

REF = "/root/reference/2d-gaussian-splatting"
_FACTORIES = ("zeros", "ones", "empty", "full", "tensor", "arange", "zeros_like", "ones_like", "empty_like", "full_like",
              "rand", "randn", "eye", "linspace", "as_tensor")


def _cpu_device(fn):
    def wrapped(*a, **k):
        if "device" in k and str(k["device"]).startswith("cuda"):
            k["device"] = "cpu"
        return fn(*a, **k)
    return wrapped


@contextlib.contextmanager
def reference_modules(stubs=None):
    """Context manager: inside it `import utils.point_utils`, `import scene.gaussian_model`,
    `import gaussian_renderer` resolve to the reference's files.  `stubs`: extra {module name: module} entries."""
    saved_modules = dict(sys.modules)
    saved_path = list(sys.path)
    saved = {n: getattr(torch, n) for n in _FACTORIES}
    saved_cuda = torch.Tensor.cuda
    try:
        for name in ("cv2", "matplotlib", "matplotlib.pyplot", "plyfile", "simple_knn", "simple_knn._C", "open3d", "trimesh",
                     "tqdm_stub"):
            if name not in sys.modules:
                sys.modules[name] = types.ModuleType(name)
        sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
        sys.modules["simple_knn._C"].distCUDA2 = None
        sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
        scene = types.ModuleType("scene")
        scene.__path__ = [REF + "/scene"]
        sys.modules["scene"] = scene
        for k, v in (stubs or {}).items():
            sys.modules[k] = v
        sys.path.insert(0, REF)
        for n in _FACTORIES:
            setattr(torch, n, _cpu_device(saved[n]))
        torch.Tensor.cuda = lambda self, *a, **k: self
        yield
    finally:
        torch.Tensor.cuda = saved_cuda
        for n, f in saved.items():
            setattr(torch, n, f)
        sys.path[:] = saved_path
        for k in list(sys.modules):
            if k not in saved_modules:
                del sys.modules[k]
