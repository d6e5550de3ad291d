"""`simple_knn._C` for MI355X (knn/ext.cpp:15-17, knn/spatial.cu:15-26).

    distCUDA2(points[P,3]) -> float32[P]   mean squared distance to the 3 nearest other points

Unlike the reference, the native side allocates nothing: the scratch it needs is a torch
tensor sized by g4s_knn_workspace()."""
import ctypes

import torch

from .. import _lib


def distCUDA2(points):
    if not points.is_cuda:
        raise RuntimeError("points must be a CUDA tensor")
    lib = _lib.load()
    P = int(points.size(0))
    dev = points.device
    means = torch.zeros((P,), dtype=torch.float32, device=dev)  # torch::full({P}, 0.0), spatial.cu:21
    if P == 0:
        return means
    with torch.cuda.device(dev):
        pts = points.float().contiguous()
        nbytes = lib.g4s_knn_workspace(P)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = lib.g4s_knn_mean_dist(P, ctypes.c_void_p(pts.data_ptr()), ctypes.c_void_p(means.data_ptr()),
                                   ctypes.c_void_p(ws.data_ptr()), nbytes, stream)
        if rc != 0:
            raise RuntimeError(f"distCUDA2 failed ({rc}): {_lib.last_error()}")
    return means
