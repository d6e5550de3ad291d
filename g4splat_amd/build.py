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


if __name__ == "__main__":
    print(build(force=True, verbose=True))
