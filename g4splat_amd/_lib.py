"""ctypes loader of the C-ABI library (include/g4s_rasterizer.h, include/g4s_render_maps.h).

There is deliberately NO fallback: if libg4s_hip.so is missing or fails to load, importing the
operators raises.  The product path never routes through the CPU oracle or any torch emulation."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libg4s_hip.so")

RESIZE_FN = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)

c_f = ctypes.c_float
c_i = ctypes.c_int
c_p = ctypes.c_void_p
c_sz = ctypes.c_size_t


class G4sPackedRows(ctypes.Structure):  # g4s_packed_rows
    _fields_ = [("rows", ctypes.c_void_p), ("block_offs", ctypes.c_void_p), ("capacity", ctypes.c_longlong)]


class G4sLayout(ctypes.Structure):
    _fields_ = [(n, ctypes.c_size_t) for n in (
        "rec", "clamped", "depth_sorted", "tiles_touched", "geom_bytes", "entries", "qhit", "binning_bytes", "ranges",
        "final_T", "n_contrib", "tile_order", "image_bytes", "hot_count")]


# symbol -> (restype, argtypes); tests/test_abi.py checks this table against include/g4s_rasterizer.h
SIGNATURES = {
    "g4s_last_error": (ctypes.c_char_p, []),
    "g4s_version": (ctypes.c_char_p, []),
    "g4s_rasterizer_forward": (c_i, [RESIZE_FN, c_p, RESIZE_FN, c_p, RESIZE_FN, c_p, c_i, c_i, c_i, c_p, c_i, c_i,
                                     c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_i, c_p, c_p,
                                     c_p, c_i, c_p]),
    "g4s_rasterizer_backward_workspace": (c_sz, [c_i, c_i]),
    "g4s_rasterizer_backward": (c_i, [c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p,
                                      c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                      c_p, c_p, c_p, c_sz, c_i, c_p]),
    "g4s_rasterizer_forward_split_sh": (c_i, [RESIZE_FN, c_p, RESIZE_FN, c_p, RESIZE_FN, c_p, c_i, c_i, c_i, c_p, c_i,
                                              c_i, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_i,
                                              c_p, c_p, c_p, c_i, c_p]),
    "g4s_rasterizer_backward_split_sh": (c_i, [c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p,
                                               c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                               c_p, c_p, c_p, c_p, c_p, c_p, c_sz, c_i, c_p]),
    "g4s_rasterizer_backward_accumulate": (c_i, [c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p,
                                                 c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                                 c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_sz, c_p, c_i, c_p]),
    "g4s_rasterizer_backward_accumulate_packed": (c_i, [c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p,
                                                        c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                                        c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_sz, c_p, c_i, c_p]),
    "g4s_rasterizer_forward_presized": (c_i, [c_p, c_sz, c_p, c_sz, c_p, c_sz, c_i, c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_p,
                                              c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_i,
                                              c_p]),
    "g4s_rasterizer_mark_visible": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p]),
    "g4s_knn_workspace": (c_sz, [c_i]),
    "g4s_knn_mean_dist": (c_i, [c_i, c_p, c_p, c_p, c_sz, c_p]),
    "g4s_rasterizer_layout": (c_i, [c_i, c_i, c_i, c_i, ctypes.POINTER(G4sLayout)]),
    "g4s_set_option": (c_i, [ctypes.c_char_p, c_i]),
    "g4s_get_option": (c_i, [ctypes.c_char_p, ctypes.POINTER(c_i)]),
    "g4s_pack_rows": (c_i, [c_i, c_p, c_p, c_p, c_i, c_p, c_i, c_p]),
    "g4s_accumulate_rows": (c_i, [c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_p]),
    "g4s_accumulate_rows_ordered": (c_i, [c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "g4s_profile_enable": (None, [c_i]),
    "g4s_profile_kernels": (c_i, []),
    "g4s_profile_name": (ctypes.c_char_p, [c_i]),
    "g4s_profile_read": (c_i, [c_i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i)]),
    "g4s_profile_reset": (None, []),
    # include/g4s_optim.h
    "g4s_adam_step": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_p]),
    "g4s_adam_step_device": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                   c_p]),
    "g4s_densify_stats": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "g4s_activations_forward": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "g4s_activations_backward": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "g4s_compact_workspace": (c_sz, [c_i]),
    "g4s_compact_scan": (c_i, [c_i, c_p, c_p, c_p, c_sz, c_p]),
    "g4s_compact_gather": (c_i, [c_i, c_p, c_p, c_i, c_p, c_p, c_p, ctypes.c_longlong, c_p]),
    # include/g4s_losses.h
    "g4s_photometric_workspace": (c_sz, [c_i, c_i]),
    "g4s_photometric_loss": (c_i, [c_i, c_i, c_p, c_p, c_f, c_p, c_p, c_p, c_sz, c_p]),
    "g4s_geometry_regularizers_workspace": (c_sz, [c_i, c_i]),
    "g4s_geometry_regularizers_forward": (c_i, [c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "g4s_geometry_regularizers_backward": (c_i, [c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    # include/g4s_render_maps.h
    "g4s_render_maps_workspace": (c_sz, []),
    "g4s_render_maps_forward": (c_i, [c_i, c_i, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_sz,
                                      c_p]),
    "g4s_render_maps_backward": (c_i, [c_i, c_i, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                       c_p, c_sz, c_p]),
}

_lib = None


def load():
    """Load libg4s_hip.so and bind every exported symbol.  Raises RuntimeError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch-ROCm bundles its own HIP runtime (same SONAME as /opt/rocm's).  It must be the one
    # already resident when libg4s_hip.so is mapped, otherwise two runtimes end up in the process
    # and torch's stream / memory handles are meaningless to ours.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m g4splat_amd.build` "
            "(or __graft_entry__.build()); there is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().g4s_last_error().decode()


OPTION_UNSET = -2147483648  # G4S_OPTION_UNSET


def set_option(name, value):
    """Diagnostic switch of the library (include/g4s_rasterizer.h: g4s_set_option).  The library never reads the
    environment on a call path; the parity tests flip these instead."""
    if load().g4s_set_option(name.encode(), int(value)) != 0:
        raise ValueError(last_error())


def get_option(name):
    v = c_i(0)
    if load().g4s_get_option(name.encode(), ctypes.byref(v)) != 0:
        raise ValueError(last_error())
    return v.value


class option:
    """`with _lib.option("no_fastpath", 1): ...` -- sets a diagnostic switch and restores its previous value."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.prev = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.prev)
        return False
