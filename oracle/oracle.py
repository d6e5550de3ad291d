"""ctypes/numpy front-end of the CPU oracle (oracle/surfel_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by anything under g4splat_amd/.  PARITY UNPINNED (see the header
of surfel_oracle.c): the reference has no tests/golden vectors for this path and cannot be
built or run in this project.

The call surface mirrors the reference's `_C` module
(dsr/rasterize_points.h:18-68, knn/spatial.h:14) on numpy arrays:

    Oracle().rasterize_gaussians(bg, means3D, colors, opacity, scales, rotations,
        scale_modifier, transMat_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy,
        H, W, sh, degree, campos, prefiltered) -> (R, color, others, radii)
    Oracle().rasterize_gaussians_backward(dL_dout_color, dL_dout_others)
        -> dict(means2D, colors, opacity, means3D, transMat, sh, scales, rotations, normal)
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsurfel_oracle.so")
_lib = None


def build(force=False):
    """Compile the C restatement with the committed Makefile (gcc only)."""
    src = os.path.join(_HERE, "surfel_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "libsurfel_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.oracle_state_new.restype = ctypes.c_void_p
        L.oracle_state_free.argtypes = [ctypes.c_void_p]
        L.oracle_forward.restype = ctypes.c_int
        L.oracle_state_R.restype = ctypes.c_int
        L.oracle_state_R.argtypes = [ctypes.c_void_p]
        L.oracle_get_higher_msb.restype = ctypes.c_uint32
        L.oracle_get_higher_msb.argtypes = [ctypes.c_uint32]
        for name in ("depths", "clamped", "means2D", "transMat", "normal_opacity", "rgb", "tiles_touched",
                     "point_offsets", "keys", "point_list", "ranges", "final_T", "n_contrib",
                     "dL_dtransMat_raw", "dL_dnormal3D", "dL_dmean2D_raw"):
            f = getattr(L, "oracle_state_" + name)
            f.restype = ctypes.c_void_p
            f.argtypes = [ctypes.c_void_p]
        _lib = L
    return _lib


def _f32(a):
    if a is None:
        return None
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _ptr(a):
    if a is None or a.size == 0:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(a.ctypes.data)


class Oracle:
    """One forward (+ optional backward) of the restated reference rasterizer on the CPU."""

    def __init__(self):
        self._L = lib()
        self._s = ctypes.c_void_p(self._L.oracle_state_new())
        self._fw = None

    def __del__(self):
        try:
            self._L.oracle_state_free(self._s)
        except Exception:
            pass

    # -- dsr/rasterize_points.cu:39-134 -------------------------------------------------
    def rasterize_gaussians(self, bg, means3D, colors, opacity, scales, rotations, scale_modifier,
                            transMat_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height,
                            image_width, sh, degree, campos, prefiltered=False):
        means3D = _f32(means3D)
        if means3D.ndim != 2 or means3D.shape[1] != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")
        P = means3D.shape[0]
        H, W = int(image_height), int(image_width)
        a = dict(bg=_f32(bg), means3D=means3D, colors=_f32(colors), opacity=_f32(opacity), scales=_f32(scales),
                 rotations=_f32(rotations), transMat=_f32(transMat_precomp), view=_f32(viewmatrix),
                 proj=_f32(projmatrix), sh=_f32(sh), campos=_f32(campos))
        M = 0
        if a["sh"] is not None and a["sh"].size != 0:
            M = a["sh"].shape[1]
        color = np.zeros((3, H, W), np.float32)
        others = np.zeros((7, H, W), np.float32)
        radii = np.zeros((P,), np.int32)
        R = self._L.oracle_forward(
            self._s, ctypes.c_int(P), ctypes.c_int(int(degree)), ctypes.c_int(M), _ptr(a["bg"]),
            ctypes.c_int(W), ctypes.c_int(H), _ptr(means3D), _ptr(a["sh"]), _ptr(a["colors"]),
            _ptr(a["opacity"]), _ptr(a["scales"]), ctypes.c_float(scale_modifier), _ptr(a["rotations"]),
            _ptr(a["transMat"]), _ptr(a["view"]), _ptr(a["proj"]), _ptr(a["campos"]),
            ctypes.c_float(tan_fovx), ctypes.c_float(tan_fovy), ctypes.c_int(int(prefiltered)),
            _ptr(color), _ptr(others), _ptr(radii))
        self._fw = dict(a=a, P=P, M=M, D=int(degree), H=H, W=W, sm=float(scale_modifier),
                        tx=float(tan_fovx), ty=float(tan_fovy), R=R)
        return R, color, others, radii

    # -- dsr/rasterize_points.cu:136-233 ------------------------------------------------
    def rasterize_gaussians_backward(self, dL_dout_color, dL_dout_others):
        fw = self._fw
        a, P, M = fw["a"], fw["P"], fw["M"]
        dpix, ddep = _f32(dL_dout_color), _f32(dL_dout_others)
        g = dict(means2D=np.zeros((P, 3), np.float32), normal=np.zeros((P, 3), np.float32),
                 opacity=np.zeros((P, 1), np.float32), colors=np.zeros((P, 3), np.float32),
                 means3D=np.zeros((P, 3), np.float32), transMat=np.zeros((P, 9), np.float32),
                 sh=np.zeros((P, M, 3), np.float32), scales=np.zeros((P, 2), np.float32),
                 rotations=np.zeros((P, 4), np.float32))
        self._L.oracle_backward(
            self._s, ctypes.c_int(P), ctypes.c_int(fw["D"]), ctypes.c_int(M), _ptr(a["bg"]),
            ctypes.c_int(fw["W"]), ctypes.c_int(fw["H"]), _ptr(a["means3D"]), _ptr(a["sh"]), _ptr(a["colors"]),
            _ptr(a["scales"]), ctypes.c_float(fw["sm"]), _ptr(a["rotations"]), _ptr(a["transMat"]),
            _ptr(a["view"]), _ptr(a["proj"]), _ptr(a["campos"]), ctypes.c_float(fw["tx"]),
            ctypes.c_float(fw["ty"]), _ptr(dpix), _ptr(ddep), _ptr(g["means2D"]), _ptr(g["normal"]),
            _ptr(g["opacity"]), _ptr(g["colors"]), _ptr(g["means3D"]), _ptr(g["transMat"]), _ptr(g["sh"]),
            _ptr(g["scales"]), _ptr(g["rotations"]))
        return g

    # -- intermediate state (stage-wise comparison with the HIP path) ---------------------
    def state(self, name):
        fw = self._fw
        P, N, R = fw["P"], fw["H"] * fw["W"], fw["R"]
        tiles = ((fw["W"] + 15) // 16) * ((fw["H"] + 15) // 16)
        spec = dict(depths=(np.float32, (P,)), clamped=(np.uint8, (P, 3)), means2D=(np.float32, (P, 2)),
                    transMat=(np.float32, (P, 9)), normal_opacity=(np.float32, (P, 4)),
                    rgb=(np.float32, (P, 3)), tiles_touched=(np.uint32, (P,)), point_offsets=(np.uint32, (P,)),
                    keys=(np.uint64, (R,)), point_list=(np.uint32, (R,)), ranges=(np.uint32, (tiles, 2)),
                    final_T=(np.float32, (3, N)), n_contrib=(np.uint32, (2, N)),
                    dL_dtransMat_raw=(np.float32, (P, 9)), dL_dnormal3D=(np.float32, (P, 3)), dL_dmean2D_raw=(np.float32, (P, 3)))
        dt, shape = spec[name]
        n = int(np.prod(shape))
        if n == 0:
            return np.zeros(shape, dt)
        p = getattr(self._L, "oracle_state_" + name)(self._s)
        buf = (ctypes.c_char * (n * np.dtype(dt).itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=dt).reshape(shape).copy()


def instance_flags(o):
    """uint8[R]: 1 where the (tile, Gaussian) instance of Oracle `o` has a pixel passing the alpha test."""
    R = o._fw["R"]
    flags = np.zeros((max(R, 1),), np.uint8)
    lib().oracle_instance_flags(o._s, _ptr(flags))
    return flags[:R]


def lane_model(o):
    """-> dict of the blend backward's lane-utilisation model for Oracle `o`'s forward (oracle_lane_model in
    surfel_oracle.c; tools/lane_util_model.py prints it)."""
    out = np.zeros(15, np.float64)
    lib().oracle_lane_model(o._s, ctypes.c_void_p(out.ctypes.data))
    names = ("entries", "pairs", "visits_quadrants", "visits_row_strips", "visits_col_strips", "visits_free_halves",
             "visits_free_cells", "visits_lower_bound", "visits_saved_by_entry_pairing", "visits_in_one_half",
             "visits_in_one_row", "visits_le16_lanes", "visits_le8_lanes", "pairs_in_le16", "pairs_in_le8")
    return dict(zip(names, out.tolist()))


def pixel_margins(o):
    """float32[3, H*W]: per pixel, the smallest relative distance to a threshold met by each kind of discrete
    decision of the forward loop (0: skip a splat -- alpha vs 1/255, depth vs near; 1: stop the pixel -- T(1-alpha)
    vs 1e-4; 2: median depth -- T vs 0.5).  See oracle_pixel_margins in surfel_oracle.c."""
    fw = o._fw
    N = fw["H"] * fw["W"]
    out = np.full((3, N), np.finfo(np.float32).max, np.float32)
    if fw["P"] == 0:
        return out
    tm = fw["a"]["transMat"]
    lib().oracle_pixel_margins(o._s, _ptr(tm if tm is not None and tm.size else None), _ptr(out))
    return out


def skip_suspects(o, margin, with_margins=False):
    """(int64[n] flat pixel indices, int32[n] Gaussian ids): the (pixel, Gaussian) pairs of Oracle `o`'s forward whose skip
    decision (alpha vs 1/255, depth vs near) sits within `margin` of its threshold and is reached before the pixel stops.
    with_margins: also pixel_margins(o), from the same walk over the frame.  See oracle_skip_suspects in surfel_oracle.c."""
    fw = o._fw
    N = fw["H"] * fw["W"]
    margins = np.full((3, N), np.finfo(np.float32).max, np.float32) if with_margins else None
    if fw["P"] == 0:
        empty = (np.zeros(0, np.int64), np.zeros(0, np.int32))
        return empty + (margins,) if with_margins else empty
    tm = fw["a"]["transMat"]
    tmp = _ptr(tm if tm is not None and tm.size else None)
    L = lib()
    L.oracle_skip_suspects.restype = ctypes.c_long
    cap = 1 << 16
    while True:
        pix, gid = np.zeros(cap, np.int64), np.zeros(cap, np.int32)
        n = int(L.oracle_skip_suspects(o._s, tmp, ctypes.c_float(margin), ctypes.c_long(cap),
                                       ctypes.c_void_p(pix.ctypes.data), ctypes.c_void_p(gid.ctypes.data), _ptr(margins)))
        if n <= cap:
            return (pix[:n], gid[:n], margins) if with_margins else (pix[:n], gid[:n])
        cap = n


def pixel_alternatives(o, pix_ids, margin):
    """float64[len(pix_ids), ALT, 12]: the outputs of the listed pixels (flat indices y * W + x) with their decisions
    that sit within `margin` of a threshold taken the other way, one combination per ALT slot (slot 0 = as rendered).
    Columns: colour 0..2, the seven `others` maps 3..9, Gaussian id of the last / median contributor 10, 11 (-1 = none;
    exact in float32 below 2^24 Gaussians).  See oracle_pixel_alternatives in surfel_oracle.c."""
    fw = o._fw
    ids = np.ascontiguousarray(np.asarray(pix_ids, dtype=np.int64))
    L = lib()
    L.oracle_pixel_alt_count.restype = ctypes.c_int
    alt = int(L.oracle_pixel_alt_count())
    out = np.zeros((len(ids), alt, 12), np.float32)
    if len(ids) == 0 or fw["P"] == 0:
        return out.astype(np.float64)
    tm, col = fw["a"]["transMat"], fw["a"]["colors"]
    L.oracle_pixel_alternatives(o._s, _ptr(col if col is not None and col.size else None),
                                _ptr(tm if tm is not None and tm.size else None), _ptr(fw["a"]["bg"]),
                                ctypes.c_float(margin), ctypes.c_int(len(ids)), ctypes.c_void_p(ids.ctypes.data), _ptr(out))
    return out.astype(np.float64)


def mark_visible(means3D, viewmatrix, projmatrix):
    """dsr/rasterize_points.cu:235-254"""
    m = _f32(means3D)
    out = np.zeros((m.shape[0],), np.uint8)
    lib().oracle_mark_visible(ctypes.c_int(m.shape[0]), _ptr(m), _ptr(_f32(viewmatrix)), _ptr(_f32(projmatrix)),
                              _ptr(out))
    return out.astype(bool)


def distCUDA2(points):
    """knn/spatial.cu:15-26"""
    p = _f32(points)
    out = np.zeros((p.shape[0],), np.float32)
    lib().oracle_knn(ctypes.c_int(p.shape[0]), _ptr(p), _ptr(out))
    return out


def distCUDA2_queries(points, queries):
    """distCUDA2 restricted to the rows `queries` (all points are neighbour candidates): brute force, O(len(queries) P)."""
    p = _f32(points)
    q = np.ascontiguousarray(np.asarray(queries, dtype=np.int32))
    out = np.zeros((q.shape[0],), np.float32)
    lib().oracle_knn_queries(ctypes.c_int(p.shape[0]), _ptr(p), ctypes.c_int(q.shape[0]), ctypes.c_void_p(q.ctypes.data), _ptr(out))
    return out


def get_higher_msb(n):
    return int(lib().oracle_get_higher_msb(ctypes.c_uint32(n)))
