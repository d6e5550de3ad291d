#!/usr/bin/env python
"""Which gradient rows differ most from the oracle's, and what kind of splat they belong to (run on the GPU box):
    python tools/row_debug.py <view> [tensor] [no_fastpath]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np

from common import cotangents, hip_state, parity_report, run_hip, run_oracle
from g4splat_amd import _lib
from oracle import oracle as om
from parity_report import room_inputs

view = int(sys.argv[1]) if len(sys.argv) > 1 else 7
name = sys.argv[2] if len(sys.argv) > 2 else "scales"
if "no_fastpath" in sys.argv:
    _lib.set_option("no_fastpath", 1)
inp = room_inputs(1_500_000, 1600, 1200, view, 8)
g = cotangents(1200, 1600, seed=3)
o = run_oracle(om, inp, g)
h = run_hip(inp, g)
st = hip_state(h, inp)
a = h["grads"][name].astype(np.float64).reshape(len(h["grads"][name]), -1)
b = o["grads"][name].astype(np.float64).reshape(len(a), -1)
scale = np.abs(b).max()
row_err = np.abs(a - b).max(axis=1)
row_mag = np.abs(b).max(axis=1)
big = row_mag > 1e-3 * scale
rel = np.where(big, row_err / np.maximum(row_mag, 1e-300), 0.0)
orc = o["oracle"]
T = orc.state("transMat")
for i in np.argsort(-rel)[:8]:
    aff = int(st["rec_u32"][i, 3] >> 31)
    ext = st["rec_u32"][i, 3] & 0x7FFFFFFF
    print(f"row {i}: rel {rel[i]:.3g} mag {row_mag[i] / scale:.3g} of max; affine {aff}; radius {h['radii'][i]}; rect {ext & 0xFFFF}x{ext >> 16}; "
          f"centre ({st['rec'][i, 0]:.1f}, {st['rec'][i, 1]:.1f}); opacity {st['rec'][i, 7]:.3f}; scales {inp['scales'][i]}")
    print("   hip ", a[i], " oracle ", b[i])
    print("   T ", T[i])
    for other in ("transMat", "means3D", "rotations", "opacity"):
        x = h["grads"][other].reshape(len(a), -1)[i].astype(np.float64)
        y = o["grads"][other].reshape(len(a), -1)[i].astype(np.float64)
        print(f"   {other}: max rel diff {np.abs(x - y).max() / (np.abs(y).max() + 1e-300):.3g}  |oracle| {np.abs(y).max():.3g}")
