"""The drop-in boundary used from plain C (tests/hip_unit/abi_client.c: HIP runtime C API + include/g4s_rasterizer.h
+ libg4s_hip.so, no torch / Python / C++ on that side, its own non-default stream and hipMalloc-backed resize
callbacks): outputs and gradients must match the CPU oracle exactly like the Python front-end's."""
import os
import subprocess

import numpy as np
import pytest

from common import cotangents, rel_err, run_oracle, scene_inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_client_matches_oracle(hip_lib, oracle_mod, tmp_path):
    exe = str(tmp_path / "abi_client")
    subprocess.run(["gcc", "-std=c99", "-O2", os.path.join(ROOT, "tests", "hip_unit", "abi_client.c"),
                    "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "g4splat_amd"),
                    "-lg4s_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "g4splat_amd"),
                    "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    inp = scene_inputs(P=5000, W=208, H=120, seed=21, D=2, bg=(0.2, 0.3, 0.1), scale_mul=1.5)
    inp["scale_modifier"] = 1.0
    P, W, H, M = 5000, 208, 120, inp["sh"].shape[1]
    gc, go = cotangents(H, W, seed=4)
    f32 = lambda a: np.ascontiguousarray(a, np.float32).tobytes()
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(np.array([P, inp["D"], M, W, H], np.int32).tobytes())
        f.write(np.array([inp["tanfovx"], inp["tanfovy"], 1.0], np.float32).tobytes())
        for a in (inp["bg"], inp["means3D"], inp["sh"], inp["opacity"], inp["scales"], inp["rotations"], inp["view"],
                  inp["proj"], inp["campos"], gc, go):
            f.write(f32(a))
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    res = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=300, env=env)
    assert res.returncode == 0 and "abi_client ok" in res.stdout, res.stdout
    raw = open(tmp_path / "out.bin", "rb").read()
    off = [0]

    def take(dtype, *shape):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        a = np.frombuffer(raw[off[0]:off[0] + n], dtype).reshape(shape)
        off[0] += n
        return a

    R = int(take(np.int32, 1)[0])
    color, others, radii = take(np.float32, 3, H, W), take(np.float32, 7, H, W), take(np.int32, P)
    g = dict(means2D=take(np.float32, P, 3), opacity=take(np.float32, P, 1), means3D=take(np.float32, P, 3),
             sh=take(np.float32, P, M, 3), scales=take(np.float32, P, 2), rotations=take(np.float32, P, 4))
    present, knn = take(np.uint8, P), take(np.float32, P)
    assert off[0] == len(raw)
    o = run_oracle(oracle_mod, inp, (gc, go))
    assert R == o["R"]
    np.testing.assert_array_equal(radii, o["radii"])
    assert np.abs(color - o["color"]).max() <= 1e-4 and np.abs(others - o["others"]).max() <= 1e-4
    for k, v in g.items():
        assert rel_err(v, o["grads"][k]) <= 1e-3, k
    np.testing.assert_array_equal(present.astype(bool), oracle_mod.mark_visible(inp["means3D"], inp["view"], inp["proj"]))
    np.testing.assert_array_equal(knn, oracle_mod.distCUDA2(inp["means3D"]))
