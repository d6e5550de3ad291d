"""Runs the standalone wave64-primitive check (tests/hip_unit/wave_ops.hip) on the GPU box."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_wave_primitives(tmp_path):
    src = os.path.join(HERE, "hip_unit", "wave_ops.hip")
    exe = str(tmp_path / "wave_ops")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                    "-Wno-unused-value", src, "-o", exe], check=True)
    res = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert res.returncode == 0 and "wave_ops OK" in res.stdout, res.stdout
