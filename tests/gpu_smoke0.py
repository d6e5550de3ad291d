"""Bring-up script: exercises the simplest kernels first, with G4S_TRACE staging."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
print("torch", torch.__version__, "hip", torch.version.hip, "dev", torch.cuda.get_device_name(0), flush=True)
from g4splat_amd import _lib
L = _lib.load(); print(L.g4s_version(), flush=True)
from g4splat_amd.diff_surfel_rasterization import _C
from common import scene_inputs
from oracle import oracle as om
inp = scene_inputs(P=5000, seed=3)
got = _C.mark_visible(torch.as_tensor(inp["means3D"], device="cuda"), torch.as_tensor(inp["view"], device="cuda"), torch.as_tensor(inp["proj"], device="cuda")).cpu().numpy()
print("mark_visible ok", np.array_equal(got, om.mark_visible(inp["means3D"], inp["view"], inp["proj"])), flush=True)
from g4splat_amd.simple_knn._C import distCUDA2
pts = np.random.default_rng(0).normal(size=(3000, 3)).astype(np.float32)
d = distCUDA2(torch.as_tensor(pts, device="cuda")).cpu().numpy()
w = om.distCUDA2(pts)
print("knn equal", np.array_equal(d, w), np.abs(d - w).max(), flush=True)
