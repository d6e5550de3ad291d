"""The C-ABI library loads without a GPU and exports exactly what include/*.h declare;
the ctypes signature table of the Python front-end matches the header (no compute calls here)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = [os.path.join(ROOT, "include", h) for h in ("g4s_rasterizer.h", "g4s_render_maps.h", "g4s_losses.h", "g4s_optim.h")]


def declared_functions():
    src = "\n".join(open(h).read() for h in HEADERS)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef\s+struct.*?\}\s*\w+\s*;", "", src, flags=re.S)
    src = re.sub(r"typedef[^;]*;", "", src)
    decls = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(g4s_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        args = [a.strip() for a in m.group(3).replace("\n", " ").split(",")]
        if args == ["void"] or args == [""]:
            args = []
        decls[m.group(2)] = (m.group(1).strip(), args)
    return decls


def test_library_exports_every_declared_symbol(hip_lib):
    from g4splat_amd import _lib
    decls = declared_functions()
    assert len(decls) >= 12
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], stdout=subprocess.PIPE, text=True, check=True)
    exported = {line.split()[-1] for line in out.stdout.splitlines() if " T " in line}
    for name in decls:
        assert name in exported, f"{name} declared in the header but not exported"
        assert hasattr(hip_lib, name)
    for name in exported:
        if name.startswith("g4s_") and not name.endswith("_internal"):
            assert name in decls, f"{name} exported but not declared in include/*.h"


def test_ctypes_table_matches_header(hip_lib):
    from g4splat_amd import _lib
    decls = declared_functions()
    assert set(_lib.SIGNATURES) == set(decls)
    for name, (res, args) in _lib.SIGNATURES.items():
        assert len(args) == len(decls[name][1]), f"{name}: {len(args)} ctypes args vs {len(decls[name][1])} in header"


def test_version_and_error_strings(hip_lib):
    assert hip_lib.g4s_version().startswith(b"g4s-hip") and b"gfx950" in hip_lib.g4s_version()
    assert hip_lib.g4s_last_error() == b""
    assert hip_lib.g4s_knn_workspace(1000) > 0
    assert hip_lib.g4s_rasterizer_backward_workspace(10, 100) >= 100 * 80
    assert hip_lib.g4s_render_maps_workspace() >= 84
    # layout query is pure host code
    import ctypes
    L = _layout(hip_lib, 1000, 5000, 640, 480)
    assert L.rec == 0 and L.geom_bytes > 1000 * 96 and L.binning_bytes > 2 * 5000 * 8 and L.image_bytes > 640 * 480 * 20
    assert hip_lib.g4s_rasterizer_layout(-1, 0, 0, 0, ctypes.byref(L)) < 0
    assert b"bad layout" in hip_lib.g4s_last_error()


def test_diagnostic_switches_are_library_state_not_environment(hip_lib):
    """The test switches are integers held by the library (g4s_set_option): round trip, unknown names refused, and no
    translation unit calls getenv outside the one-time G4S_TRACE probe (nothing on a call path depends on the
    caller's environment)."""
    from g4splat_amd import _lib
    for name in ("box_only", "no_fastpath", "bwd_fwd_order", "no_side_zero"):
        assert _lib.get_option(name) == 0
        with _lib.option(name, 1):
            assert _lib.get_option(name) == 1
        assert _lib.get_option(name) == 0
    assert _lib.get_option("bwd_hot_threshold") == _lib.OPTION_UNSET
    with _lib.option("bwd_hot_threshold", 40):
        assert _lib.get_option("bwd_hot_threshold") == 40
    assert _lib.get_option("bwd_hot_threshold") == _lib.OPTION_UNSET
    with pytest.raises(ValueError, match="unknown option"):
        _lib.set_option("no_such_switch", 1)
    csrc = os.path.join(ROOT, "g4splat_amd", "csrc")
    uses = []
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            for i, line in enumerate(open(os.path.join(csrc, f)), 1):
                if "getenv" in line and not line.lstrip().startswith("//"):
                    uses.append((f, i, line.strip()))
    assert len(uses) == 1 and "G4S_TRACE" in uses[0][2] and "static" in uses[0][2], uses


def test_argument_validation_is_host_side(hip_lib):
    """Every entry point rejects bad arguments before it touches the device: negative status + a message, no launch."""
    import ctypes
    from g4splat_amd import _lib
    lib = hip_lib
    nul = ctypes.c_void_p(0)
    one = ctypes.c_void_p(256)  # never dereferenced: validation fails first
    cb = _lib.RESIZE_FN(lambda ctx, n: 0)
    INVALID = -1

    def expect(rc, text):
        assert rc == INVALID, rc
        assert text.encode() in lib.g4s_last_error(), lib.g4s_last_error()

    expect(lib.g4s_densify_stats(5, nul, one, nul, one, one, nul, nul), "NULL required pointer")
    expect(lib.g4s_densify_stats(-1, one, one, nul, one, one, nul, nul), "must not be negative")
    expect(lib.g4s_densify_stats(5, one, one, nul, one, one, one, nul), "NULL required pointer")  # max_radii2D without radii
    assert lib.g4s_densify_stats(0, nul, nul, nul, nul, nul, nul, nul) == 0
    expect(lib.g4s_geometry_regularizers_forward(0, 4, one, one, one, one, one, 1 << 20, nul), "must be positive")
    expect(lib.g4s_geometry_regularizers_forward(4, 4, one, nul, one, one, one, 1 << 20, nul), "NULL required pointer")
    expect(lib.g4s_geometry_regularizers_forward(4, 4, one, one, one, one, one, 8, nul), "workspace too small")
    expect(lib.g4s_geometry_regularizers_backward(4, 4, one, one, nul, one, one, one, nul), "NULL required pointer")
    expect(lib.g4s_photometric_loss(0, 0, one, one, 0.2, one, nul, one, 1 << 20, nul), "must be positive")
    expect(lib.g4s_adam_step(0, nul, nul, nul, nul, nul, nul, nul, 0.9, 0.999, 1e-15, nul), "1..8 segments")
    # split SH: the dc tensor and, for M > 1, the rest tensor are required
    fwd_tail = (one, one, 1.0, one, nul, one, one, one, 1.0, 1.0, 0, one, one, nul, 0, nul)
    expect(lib.g4s_rasterizer_forward_split_sh(cb, nul, cb, nul, cb, nul, 10, 3, 16, one, 32, 32, one, nul, one, *fwd_tail),
           "split SH needs")
    expect(lib.g4s_rasterizer_forward_split_sh(cb, nul, cb, nul, cb, nul, 10, 3, 16, one, 32, 32, one, one, nul, *fwd_tail),
           "split SH needs")
    expect(lib.g4s_rasterizer_forward(_lib.RESIZE_FN(), nul, cb, nul, cb, nul, 10, 3, 16, one, 32, 32, one, one, nul, *fwd_tail),
           "resize callbacks")
    expect(lib.g4s_rasterizer_forward(cb, nul, cb, nul, cb, nul, 10, 3, 16, one, 0, 32, one, one, nul, *fwd_tail),
           "must be positive")
    # the owner's accumulation of the exchange (g4s_accumulate_rows)
    segs = (ctypes.c_void_p * 2)(256, 512)
    wid = (ctypes.c_int * 2)(3, 48)
    off, cnt = (ctypes.c_int * 1)(0), (ctypes.c_int * 1)(5)
    expect(lib.g4s_accumulate_rows(0, segs, wid, 1, off, cnt, one, 0, 64, nul), "1..8 segments")
    expect(lib.g4s_accumulate_rows(2, segs, wid, 1, off, cnt, one, 64, 0, nul), "row_lo <= row_hi")
    expect(lib.g4s_accumulate_rows(2, segs, wid, 1, off, cnt, nul, 0, 64, nul), "NULL pointer")
    expect(lib.g4s_accumulate_rows(2, segs, wid, 1, (ctypes.c_int * 1)(-1), cnt, one, 0, 64, nul), "negative offset")
    assert lib.g4s_accumulate_rows(2, segs, wid, 0, nul, nul, nul, 0, 64, nul) == 0   # no source: nothing to do
    assert lib.g4s_accumulate_rows(2, segs, wid, 1, off, cnt, one, 64, 64, nul) == 0  # empty shard: nothing to do
    wide = (ctypes.c_int * 2)(200, 48)
    rc = lib.g4s_accumulate_rows(2, segs, wide, 1, off, cnt, one, 0, 64, nul)
    assert rc < 0 and b"wider than 240" in lib.g4s_last_error()


def _layout(lib, P, R, W, H):
    import ctypes
    from g4splat_amd import _lib
    L = _lib.G4sLayout()
    assert lib.g4s_rasterizer_layout(P, R, W, H, ctypes.byref(L)) == 0
    return L


def test_oracle_library_builds_and_is_not_linked_into_product(hip_lib, oracle_mod):
    from g4splat_amd import _lib
    out = subprocess.run(["ldd", _lib.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    assert "surfel_oracle" not in out
    # nothing under g4splat_amd/ may import the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "g4splat_amd")):
        for f in files:
            txt = open(os.path.join(dirpath, f), errors="ignore").read() if f.endswith((".py", ".hip", ".h")) else ""
            if f.endswith(".py"):
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "libsurfel_oracle" not in txt, f
            elif txt:
                assert not re.search(r"#\s*include\s+[\"<][^\">]*oracle", txt), f


def test_pybind_stub_of_integration_b_builds_and_exports_the_reference_surface(hip_lib):
    """INTEGRATION.md section B is compiled, not prose: g4splat_amd/pybind/{rasterize_points.cpp, ext.cpp} build as plain
    C++ (torch.utils.cpp_extension, nothing hipified) against libg4s_hip.so and export the three names of
    dsr/ext.cpp:15-19; the module links the product library and not the oracle.  (Compute: tests/test_gpu_pybind.py.)"""
    from g4splat_amd import build
    so = build.build_pybind()
    mod = build.pybind_module()
    assert {"rasterize_gaussians", "rasterize_gaussians_backward", "mark_visible"} <= set(dir(mod))
    out = subprocess.run(["ldd", so], stdout=subprocess.PIPE, text=True).stdout
    assert "libg4s_hip.so" in out and "not found" not in out.split("libg4s_hip.so")[1].splitlines()[0]
    assert "surfel_oracle" not in out
    for f in ("rasterize_points.cpp", "ext.cpp"):
        txt = open(os.path.join(ROOT, "g4splat_amd", "pybind", f)).read()
        assert "cuda_runtime" not in txt and "__global__" not in txt and "<<<" not in txt  # no kernels: plumbing only
