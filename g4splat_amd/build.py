"""Builds libg4s_hip.so (gfx950) in-tree with hipcc.  hipcc cross-compiles without a GPU, so this
also runs in the GPU-less build container; the resulting .so travels to the GPU box with the tree."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libg4s_hip.so")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")) or f == "Makefile"]
    inc = os.path.join(_HERE, "..", "include")
    srcs += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]  # every public header is a Makefile dependency
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link libg4s_hip.so.  Raises on failure."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = ["make", "-C", CSRC, "-j8"] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libg4s_hip.so failed:\n" + res.stdout[-4000:])
    return LIB_PATH


PYBIND_DIR = os.path.join(_HERE, "pybind")
PYBIND_SO = os.path.join(PYBIND_DIR, "_C_pybind.so")


def _rocm_path():
    """ROCM_PATH / ROCM_HOME if set, else the root of the hipcc on PATH, else /opt/rocm."""
    import shutil
    for k in ("ROCM_PATH", "ROCM_HOME"):
        if os.environ.get(k):
            return os.environ[k]
    hipcc = shutil.which("hipcc")
    if hipcc:
        return os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
    return "/opt/rocm"


def build_pybind(force=False, verbose=False):
    """INTEGRATION.md section B, compiled: the pybind module a maintainer of the reference would build in place of its
    CUDA extension -- pybind/rasterize_points.cpp + pybind/ext.cpp, plain C++ through torch.utils.cpp_extension (explicit
    .cpp sources: nothing is hipified), linked against libg4s_hip.so.  Built IN-TREE (pybind/_C_pybind.so travels to the
    GPU box like the library).  The product's own front-end stays the ctypes one (diff_surfel_rasterization/_C.py); this
    module exists to prove the documented route and is compared with it bit for bit (tests/test_gpu_pybind.py)."""
    srcs = [os.path.join(PYBIND_DIR, f) for f in ("rasterize_points.cpp", "ext.cpp", "rasterize_points.h")]
    deps = srcs + [LIB_PATH, os.path.join(_HERE, "..", "include", "g4s_rasterizer.h")]
    if not force and os.path.exists(PYBIND_SO) and all(os.path.getmtime(d) <= os.path.getmtime(PYBIND_SO) for d in deps):
        return PYBIND_SO
    build()
    from torch.utils import cpp_extension
    os.environ.setdefault("MAX_JOBS", "4")
    cpp_extension.load(
        name="_C_pybind", sources=srcs[:2], build_directory=PYBIND_DIR, verbose=verbose, with_cuda=False,
        is_python_module=False,  # only build here; pybind_module() below imports the file
        extra_cflags=["-O2", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1"],
        extra_include_paths=[os.path.abspath(os.path.join(_HERE, "..", "include")), os.path.join(_rocm_path(), "include")],
        extra_ldflags=[f"-L{_HERE}", "-lg4s_hip", "-Wl,-rpath,'$$ORIGIN/..'", "-lc10_hip", "-ltorch_hip"])
    if not os.path.exists(PYBIND_SO):
        raise RuntimeError("building the pybind stub failed: " + PYBIND_SO + " not produced")
    return PYBIND_SO


def pybind_module():
    """Imports the prebuilt pybind/_C_pybind.so (no compilation here: build_pybind() makes it)."""
    import importlib.util
    import torch  # noqa: F401  (libtorch must be resident before the extension is mapped)
    from . import _lib
    _lib.load()
    if not os.path.exists(PYBIND_SO):
        raise RuntimeError(f"{PYBIND_SO} not found: run g4splat_amd.build.build_pybind()")
    spec = importlib.util.spec_from_file_location("_C_pybind", PYBIND_SO)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_pybind(force=True, verbose=True))
