"""`diff_surfel_rasterization._C` for MI355X: the torch <-> raw-pointer glue of the reference
(dsr/rasterize_points.cu + dsr/ext.cpp:15-19) re-expressed over the C ABI of libg4s_hip.so.

Same three entry points, same argument order, same tuple orders, same error behaviour:

    rasterize_gaussians(...)           -> (num_rendered, color, others, radii, geomBuffer, binningBuffer, imgBuffer)
                                          dsr/rasterize_points.h:18-38, rasterize_points.cu:39-134
    rasterize_gaussians_backward(...)  -> (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dtransMat,
                                           dL_dsh, dL_dscales, dL_drotations)
                                          dsr/rasterize_points.h:40-63, rasterize_points.cu:136-233
                                          (extension: `sh` may be the pair (features_dc, features_rest); dL_dsh is
                                          then the pair of their gradients -- no concatenation on either side)
    mark_visible(means3D, viewmatrix, projmatrix) -> bool[P]      rasterize_points.cu:235-254

torch is plumbing here (device memory, current stream); all compute is in the HIP library.
There is no CPU path: tensors must live on a HIP device ("cuda" in torch-ROCm).
"""
import ctypes

import torch

from .. import _lib

NUM_CHANNELS = 3  # dsr/cuda_rasterizer/config.h:15


def _check_cuda(t, name):
    # CHECK_INPUT, rasterize_points.cu:27-28
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")


def _f32c(t, align=4):
    """contiguous float32 view whose base address satisfies `align` (clone if a view is offset)."""
    if t.dtype != torch.float32:
        t = t.float()
    t = t.contiguous()
    if t.numel() and t.data_ptr() % align:
        t = t.clone()
    return t


def _ptr(t):
    # empty tensor => NULL => "absent" (rasterizer_impl.cu:322-323)
    return ctypes.c_void_p(t.data_ptr() if t is not None and t.numel() else 0)


# Size-stable scratch.  The binning chunk and the backward's workspace grow with the frame's instance count, which differs
# from view to view; torch's caching allocator serves a request from a cached block of the SAME size class at once, but a
# stream of ever-different sizes splits and re-merges its blocks and re-enters hipMalloc every few steps (7 times in a
# 60-step pass over eight views at the metric size: the only in-step jitter the bench line still showed).  So every chunk
# is requested at the largest size this (device, P, width, height) has needed so far: after one pass over the views each
# call asks for exactly what the previous call freed.  The tensors stay per call, like the reference's (two forwards may
# be alive at once); only their SIZE is remembered.  Cost: the memory of the largest view instead of the current one.
_HIGH_WATER = {}
_SIZE_QUANTUM = 2 << 20  # the allocator's own granularity for large blocks


def _stable_size(key, nbytes):
    nbytes = int(nbytes)
    if nbytes <= 0 or key is None:
        return nbytes
    q = (nbytes + _SIZE_QUANTUM - 1) // _SIZE_QUANTUM * _SIZE_QUANTUM
    hw = _HIGH_WATER.get(key, 0)
    if q > hw:
        if len(_HIGH_WATER) > 4096:  # (a long densification schedule: every P is a new key)
            _HIGH_WATER.clear()
        _HIGH_WATER[key] = hw = q
    return hw


class _Scratch:
    """resizeFunctional (rasterize_points.cu:31-37) as a C callback.  `key`: see _stable_size.

    The callback closes over a plain dict, NOT over this object: `self -> cb -> closure -> self` would be a reference cycle,
    and the chunk tensor inside it would live until CPython's cyclic collector next runs instead of dying with the last
    reference -- 388 MiB per forward at the metric size piling up for a dozen steps, which is what made torch's allocator
    re-enter hipMalloc inside the timed region (tools/alloc_debug.py, profiles/r06_allocations.txt)."""

    def __init__(self, device, key=None):
        state = {"tensor": torch.empty(0, dtype=torch.uint8, device=device), "error": None}
        self._state = state

        def _resize(_ctx, nbytes):
            try:
                state["tensor"].resize_(_stable_size(key, nbytes))
                return state["tensor"].data_ptr() if nbytes else 1  # non-NULL sentinel for empty chunks
            except Exception as ex:  # surfaced after the C call returns G4S_ERR_ALLOC
                state["error"] = ex
                return 0

        self.cb = _lib.RESIZE_FN(_resize)

    @property
    def tensor(self):
        return self._state["tensor"]

    @property
    def error(self):
        return self._state["error"]


def _check_shapes(P, background, colors, opacity, scales, rotations, transMat_precomp, viewmatrix, projmatrix, sh, campos):
    """The native side takes raw pointers (like the reference's, which would read garbage): refuse tensors whose
    element counts cannot be what the kernels index -- e.g. 3-D Gaussian splatting's [P,3] scales."""
    def want(t, n, what):
        if t is not None and t.numel() and t.numel() != n:
            raise RuntimeError(f"{what} has {t.numel()} elements, expected {n}")
    want(background, 3, "background [3]")
    want(colors, 3 * P, "colors_precomp [P,3]")
    want(opacity, P, "opacity [P,1]")
    want(scales, 2 * P, "scales [P,2] (surfels have two scales)")
    want(rotations, 4 * P, "rotations [P,4]")
    want(transMat_precomp, 9 * P, "transMat_precomp [P,9]")
    want(viewmatrix, 16, "viewmatrix [4,4]")
    want(projmatrix, 16, "projmatrix [4,4]")
    want(campos, 3, "campos [3]")
    if sh is not None and not isinstance(sh, (tuple, list)) and sh.numel():
        if sh.ndim != 3 or sh.size(0) != P or sh.size(2) != 3:
            raise RuntimeError(f"sh must have dimensions (num_points, M, 3), got {tuple(sh.shape)}")


def _check_split_sh(sh_dc, sh_rest, means3D, colors):
    P = int(means3D.size(0))
    if sh_dc.ndim != 3 or tuple(sh_dc.shape) != (P, 1, 3) or sh_rest.ndim != 3 or sh_rest.size(0) != P or sh_rest.size(2) != 3:
        raise RuntimeError("split SH must be (features_dc [P,1,3], features_rest [P,M-1,3])")
    if colors is not None and colors.numel():
        raise RuntimeError("provide split SHs or precomputed colours, not both")


def _raise(code, what):
    raise RuntimeError(f"{what} failed ({code}): {_lib.last_error()}")


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, transMat_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug):
    if means3D.ndim != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    split = isinstance(sh, (tuple, list))  # extension: (features_dc [P,1,3], features_rest [P,M-1,3]), never concatenated
    sh_dc, sh_rest = sh if split else (sh, None)
    for name, t in (("background", background), ("means3D", means3D), ("colors", colors), ("opacity", opacity),
                    ("scales", scales), ("rotations", rotations), ("transMat_precomp", transMat_precomp),
                    ("viewmatrix", viewmatrix), ("projmatrix", projmatrix), ("sh", sh_dc), ("campos", campos)) + (
                        (("sh_rest", sh_rest),) if split else ()):
        _check_cuda(t, name)
    if split:
        _check_split_sh(sh_dc, sh_rest, means3D, colors)
    lib = _lib.load()
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    _check_shapes(P, background, colors, opacity, scales, rotations, transMat_precomp, viewmatrix, projmatrix, sh, campos)
    dev = means3D.device
    with torch.cuda.device(dev):
        fopt = dict(dtype=torch.float32, device=dev)
        key = (dev.index, P, W, H)
        geom, binning, img = _Scratch(dev, key + ("geom",)), _Scratch(dev, key + ("binning",)), _Scratch(dev, key + ("img",))
        if P == 0:  # rasterize_points.cu:85-99: zero-filled outputs, nothing launched
            return (0, torch.zeros((NUM_CHANNELS, H, W), **fopt), torch.zeros((7, H, W), **fopt),
                    torch.zeros((0,), dtype=torch.int32, device=dev), geom.tensor, binning.tensor, img.tensor)
        out_color = torch.empty((NUM_CHANNELS, H, W), **fopt)
        out_others = torch.empty((7, H, W), **fopt)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        bg, m3, col, opa = _f32c(background), _f32c(means3D), _f32c(colors), _f32c(opacity)
        sc, rot, tm = _f32c(scales, 8), _f32c(rotations, 16), _f32c(transMat_precomp)
        vm, pm, cp = _f32c(viewmatrix), _f32c(projmatrix), _f32c(campos)
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if split:
            M = 1 + int(sh_rest.size(1))
            dc, rest = _f32c(sh_dc), _f32c(sh_rest)
            rendered = lib.g4s_rasterizer_forward_split_sh(
                geom.cb, None, binning.cb, None, img.cb, None, P, int(degree), M, _ptr(bg), W, H, _ptr(m3), _ptr(dc),
                _ptr(rest), _ptr(opa), _ptr(sc), float(scale_modifier), _ptr(rot), _ptr(tm), _ptr(vm), _ptr(pm),
                _ptr(cp), float(tan_fovx), float(tan_fovy), int(bool(prefiltered)), _ptr(out_color), _ptr(out_others),
                _ptr(radii), int(bool(debug)), stream)
        else:
            M = int(sh.size(1)) if sh.size(0) != 0 else 0
            shc = _f32c(sh, 16)
            rendered = lib.g4s_rasterizer_forward(
                geom.cb, None, binning.cb, None, img.cb, None, P, int(degree), M, _ptr(bg), W, H, _ptr(m3), _ptr(shc),
                _ptr(col), _ptr(opa), _ptr(sc), float(scale_modifier), _ptr(rot), _ptr(tm), _ptr(vm), _ptr(pm),
                _ptr(cp), float(tan_fovx), float(tan_fovy), int(bool(prefiltered)), _ptr(out_color), _ptr(out_others),
                _ptr(radii), int(bool(debug)), stream)
        for s in (geom, binning, img):
            if s.error is not None:
                raise s.error
        if rendered < 0:
            _raise(rendered, "rasterize_gaussians")
    return rendered, out_color, out_others, radii, geom.tensor, binning.tensor, img.tensor


class PresizedState:
    """Caller-owned chunks for `rasterize_gaussians_presized` (extension: g4s_rasterizer_forward_presized), reusable
    from frame to frame: no allocation and no host synchronisation per forward.  `instance_capacity` bounds the
    binned (Gaussian, tile) instances of a frame; `status` is the device tensor [num_rendered, instances binned,
    emitting Gaussians, overflow flag] the library fills -- reading it (`.tolist()`) is the caller's decision."""

    def __init__(self, P, width, height, instance_capacity, device):
        lib = _lib.load()
        L = _lib.G4sLayout()
        if lib.g4s_rasterizer_layout(int(P), int(instance_capacity), int(width), int(height), ctypes.byref(L)) != 0:
            _raise(-1, "g4s_rasterizer_layout")
        self.P, self.width, self.height, self.capacity = int(P), int(width), int(height), int(instance_capacity)
        mk = lambda n: torch.empty(int(n), dtype=torch.uint8, device=device)
        self.geom, self.binning, self.img = mk(L.geom_bytes), mk(L.binning_bytes), mk(L.image_bytes)
        self.status = torch.zeros(4, dtype=torch.int32, device=device)


def rasterize_gaussians_presized(state, background, means3D, colors, opacity, scales, rotations, scale_modifier,
                                 transMat_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width,
                                 sh, degree, campos, prefiltered, debug):
    """`rasterize_gaussians` without the read-back of num_rendered: same arguments after `state` (a PresizedState), same
    outputs -- except that the first element of the returned tuple is the CAPACITY (pass it as `R` to
    rasterize_gaussians_backward); the real counts are in `state.status` on the device.  The host never waits."""
    if means3D.ndim != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    split = isinstance(sh, (tuple, list))
    sh_dc, sh_rest = sh if split else (sh, None)
    for name, t in (("background", background), ("means3D", means3D), ("colors", colors), ("opacity", opacity),
                    ("scales", scales), ("rotations", rotations), ("transMat_precomp", transMat_precomp),
                    ("viewmatrix", viewmatrix), ("projmatrix", projmatrix), ("sh", sh_dc), ("campos", campos)) + (
                        (("sh_rest", sh_rest),) if split else ()):
        _check_cuda(t, name)
    if split:
        _check_split_sh(sh_dc, sh_rest, means3D, colors)
    lib = _lib.load()
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    if (P, W, H) != (state.P, state.width, state.height):
        raise RuntimeError(f"PresizedState was built for P={state.P}, {state.width}x{state.height}")
    _check_shapes(P, background, colors, opacity, scales, rotations, transMat_precomp, viewmatrix, projmatrix, sh, campos)
    dev = means3D.device
    with torch.cuda.device(dev):
        fopt = dict(dtype=torch.float32, device=dev)
        out_color = torch.empty((NUM_CHANNELS, H, W), **fopt)
        out_others = torch.empty((7, H, W), **fopt)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        bg, m3, col, opa = _f32c(background), _f32c(means3D), _f32c(colors), _f32c(opacity)
        sc, rot, tm = _f32c(scales, 8), _f32c(rotations, 16), _f32c(transMat_precomp)
        vm, pm, cp = _f32c(viewmatrix), _f32c(projmatrix), _f32c(campos)
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if split:
            M = 1 + int(sh_rest.size(1))
            shc, rest = _f32c(sh_dc), _f32c(sh_rest)
        else:
            M = int(sh.size(1)) if sh.size(0) != 0 else 0
            shc, rest = _f32c(sh, 16), None
        rc = lib.g4s_rasterizer_forward_presized(
            _ptr(state.geom), state.geom.numel(), _ptr(state.binning), state.binning.numel(), _ptr(state.img),
            state.img.numel(), state.capacity, _ptr(state.status), P, int(degree), M, _ptr(bg), W, H, _ptr(m3), _ptr(shc),
            _ptr(rest), _ptr(col), _ptr(opa), _ptr(sc), float(scale_modifier), _ptr(rot), _ptr(tm), _ptr(vm), _ptr(pm),
            _ptr(cp), float(tan_fovx), float(tan_fovy), _ptr(out_color), _ptr(out_others), _ptr(radii), int(bool(debug)),
            stream)
        if rc < 0:
            _raise(rc, "rasterize_gaussians_presized")
    return state.capacity, out_color, out_others, radii, state.geom, state.binning, state.img


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier,
                                 transMat_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                 dL_dout_others, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug,
                                 out=None):
    """`out` (extension over the reference): optional dict name -> preallocated float32 tensor for any of the
    returned gradients, e.g. views of a persistent all-reduce bucket (parallel.py); the library writes every
    element of them, so no packing copy is needed before the collective.  `out["workspace"]`: a uint8 tensor of at
    least g4s_rasterizer_backward_workspace(P, R) bytes to use instead of a fresh allocation (a training loop keeps
    one; nothing in it needs clearing).

    `out["accumulate"] = True` (extension: g4s_rasterizer_backward_accumulate): the parameter gradients -- dL_dmeans3D,
    dL_dopacity, dL_dsh (or dL_dsh_dc / dL_dsh_rest), dL_dscales, dL_drotations, all five REQUIRED in `out` -- are added
    to what those tensors hold instead of overwriting them (rows of Gaussians this view does not see are not touched,
    nothing is zero-filled): the second and later views of a multi-view batch.  `out["accumulate"] = "first"`: the same
    entry point starting the sums (every row written, like the plain call).  `out["view_stats"]`: float32 [P,2], the
    batch's densification statistics (sum of the per-view ||dL_dmeans2D.xy||, number of views that saw the Gaussian),
    written by the "first" call and added to by the others.  `out["after"]`: a recorded torch.cuda.Event -- the stream
    waits for it between the blend backward and the accumulating per-Gaussian kernel (views in flight on several
    streams: pipeline.ViewPipeline hands out the previous view's event).  `out["packed"] = (rows, block_offs)` (only with
    accumulate = "first"; g4s_rasterizer_backward_accumulate_packed): the per-Gaussian kernel also leaves the parameter
    gradients of the visible Gaussians PACKED, one row of 3 M + 13 floats each in index order -- [dL_dmeans3D | dL_dsh |
    dL_dopacity | dL_dscales | dL_drotations | view_stats | bits(index)], the send buffer of parallel.OwnerReduce --
    in `rows` (float32 [capacity, 3 M + 13]); `block_offs` = int32 / uint32 [ceil(P / 256)], the number of Gaussians with
    radii > 0 in the 256-blocks before each block (OwnerReduce.prepack() computes both).

    The backward reads AND updates the forward's state chunks (the validity bytes of its gradient records live in the
    binning chunk): run at most one backward at a time per forward state, and with a PresizedState -- whose chunks
    are shared from frame to frame -- run the backward of a frame before the next forward into the same state."""
    split = isinstance(sh, (tuple, list))
    sh_dc, sh_rest = sh if split else (sh, None)
    for name, t in (("background", background), ("means3D", means3D), ("radii", radii), ("colors", colors),
                    ("scales", scales), ("rotations", rotations), ("transMat_precomp", transMat_precomp),
                    ("viewmatrix", viewmatrix), ("projmatrix", projmatrix), ("sh", sh_dc), ("campos", campos),
                    ("binningBuffer", binningBuffer), ("imageBuffer", imageBuffer), ("geomBuffer", geomBuffer)) + (
                        (("sh_rest", sh_rest),) if split else ()):
        _check_cuda(t, name)
    if split:
        _check_split_sh(sh_dc, sh_rest, means3D, colors)
    lib = _lib.load()
    P = int(means3D.size(0))
    _check_shapes(P, background, colors, None, scales, rotations, transMat_precomp, viewmatrix, projmatrix, sh, campos)
    if radii.numel() != P:
        raise RuntimeError(f"radii has {radii.numel()} elements, expected {P}")
    if dL_dout_color.ndim != 3 or dL_dout_color.size(0) != 3 or tuple(dL_dout_others.shape) != (7,) + tuple(dL_dout_color.shape[1:]):
        raise RuntimeError("dL_dout_color must be [3,H,W] and dL_dout_others [7,H,W]")
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    M = (1 + int(sh_rest.size(1))) if split else (int(sh.size(1)) if sh.size(0) != 0 else 0)
    dev = means3D.device
    with torch.cuda.device(dev):
        fopt = dict(dtype=torch.float32, device=dev)
        alloc = torch.zeros if P == 0 else torch.empty  # the library writes every element when P > 0
        given = dict(out) if out else {}
        accumulate = given.pop("accumulate", False)
        if accumulate not in (False, True, "first"):
            raise RuntimeError("out['accumulate'] must be True, False or 'first'")
        first_view = accumulate == "first"
        accumulate = bool(accumulate)
        after = given.pop("after", None)
        packed = given.pop("packed", None)
        view_stats = given.pop("view_stats", None)
        if view_stats is not None and not accumulate:
            raise RuntimeError("out['view_stats'] only applies to an accumulating backward (accumulate = True or 'first')")
        if view_stats is not None and (tuple(view_stats.shape) != (P, 2) or view_stats.dtype != torch.float32
                                       or view_stats.device != dev or not view_stats.is_contiguous()):
            raise RuntimeError(f"out['view_stats'] must be a contiguous float32 ({P}, 2) tensor on {dev}")
        if accumulate:
            need = ["dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations"] + (["dL_dsh_dc", "dL_dsh_rest"] if split else ["dL_dsh"])
            missing = [n for n in need if n not in given]
            if missing:
                raise RuntimeError(f"out['accumulate'] needs the running sums in out=: missing {missing}")
            if colors is not None and colors.numel():
                raise RuntimeError("out['accumulate']: colours must come from SH")
        elif after is not None:
            raise RuntimeError("out['after'] only applies to an accumulating backward")
        if packed is not None:
            if not first_view:
                raise RuntimeError("out['packed'] only applies to accumulate = 'first' (the rows hold ONE view's gradients)")
            prow, poffs = packed
            if (prow.ndim != 2 or prow.size(1) != 3 * M + 13 or prow.dtype != torch.float32 or prow.device != dev
                    or not prow.is_contiguous()):
                raise RuntimeError(f"out['packed'][0] must be a contiguous float32 (capacity, {3 * M + 13}) tensor on {dev}")
            if (poffs.numel() != (P + 255) // 256 or poffs.dtype not in (torch.int32, torch.uint32) or poffs.device != dev
                    or not poffs.is_contiguous()):
                raise RuntimeError(f"out['packed'][1] must be a contiguous int32 ({(P + 255) // 256},) tensor on {dev}")

        def make(shape, name=None, align=4, **kw):
            t = given.pop(name, None)
            if t is None:
                return alloc(shape, **kw)
            if (tuple(t.shape) != tuple(shape) or t.dtype != torch.float32 or t.device != dev
                    or not t.is_contiguous() or (t.numel() and t.data_ptr() % align)):
                raise RuntimeError(f"out['{name}'] must be a contiguous float32 {tuple(shape)} tensor on {dev}, "
                                   f"{align}-byte aligned")
            if P == 0 and (first_view or not accumulate):
                t.zero_()
            return t

        dL_dmeans3D = make((P, 3), "dL_dmeans3D", **fopt)
        dL_dmeans2D = make((P, 3), "dL_dmeans2D", **fopt)
        dL_dcolors = make((P, NUM_CHANNELS), "dL_dcolors", **fopt)
        dL_dnormal = None  # scratch of the reference's K7 -> K8 hand-over (rasterize_points.cu:176), never returned: not written
        dL_dopacity = make((P, 1), "dL_dopacity", **fopt)
        # out["dL_dtransMat"] = False: the caller does not want it (the autograd node when no T matrices were given: the
        # reference computes dL_dtransMat as an intermediate there and nobody reads it) -- 36 B per Gaussian K8 need not write
        if given.get("dL_dtransMat", None) is False:
            given.pop("dL_dtransMat")
            dL_dtransMat = None
        else:
            dL_dtransMat = make((P, 9), "dL_dtransMat", **fopt)
        if split:
            dL_dsh = (make((P, 1, 3), "dL_dsh_dc", **fopt), make((P, M - 1, 3), "dL_dsh_rest", **fopt))
        else:
            dL_dsh = make((P, M, 3), "dL_dsh", **fopt)
        dL_dscales = make((P, 2), "dL_dscales", 8, **fopt)
        dL_drotations = make((P, 4), "dL_drotations", 16, **fopt)
        workspace = given.pop("workspace", None)
        if given:
            raise RuntimeError(f"unknown out= entries: {sorted(given)}")
        if P != 0:
            ws_bytes = lib.g4s_rasterizer_backward_workspace(P, int(R))
            if workspace is None:
                workspace = torch.empty(_stable_size((dev.index, P, W, H, "bwd"), ws_bytes), dtype=torch.uint8, device=dev)
            elif (workspace.dtype != torch.uint8 or workspace.device != dev or not workspace.is_contiguous()
                  or workspace.numel() < ws_bytes):
                raise RuntimeError(f"out['workspace'] must be a contiguous uint8 tensor of >= {ws_bytes} bytes on {dev}")
            ws_bytes = int(workspace.numel())
            bg, m3, col = _f32c(background), _f32c(means3D), _f32c(colors)
            sc, rot, tm = _f32c(scales, 8), _f32c(rotations, 16), _f32c(transMat_precomp)
            vm, pm, cp = _f32c(viewmatrix), _f32c(projmatrix), _f32c(campos)
            gc, go = _f32c(dL_dout_color), _f32c(dL_dout_others)
            rad = radii.contiguous()
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            if accumulate:
                dc, rest = (_f32c(sh_dc), _f32c(sh_rest)) if split else (_f32c(sh, 16), None)
                gdc, grest = (dL_dsh[0], dL_dsh[1]) if split else (dL_dsh, None)
                ev = ctypes.c_void_p(after.cuda_event if after is not None else 0)
                pk = None
                if packed is not None:
                    pk = _lib.G4sPackedRows(packed[0].data_ptr(), packed[1].data_ptr(), int(packed[0].size(0)))
                rc = lib.g4s_rasterizer_backward_accumulate_packed(
                    P, int(degree), M, int(R), _ptr(bg), W, H, _ptr(m3), _ptr(dc), _ptr(rest), _ptr(sc),
                    float(scale_modifier), _ptr(rot), _ptr(tm), _ptr(vm), _ptr(pm), _ptr(cp), float(tan_fovx),
                    float(tan_fovy), _ptr(rad), _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer), _ptr(gc),
                    _ptr(go), _ptr(dL_dmeans2D), _ptr(dL_dnormal), _ptr(dL_dopacity), _ptr(dL_dcolors),
                    _ptr(dL_dmeans3D), _ptr(dL_dtransMat), _ptr(gdc), _ptr(grest), _ptr(dL_dscales),
                    _ptr(dL_drotations), _ptr(view_stats), int(first_view),
                    ctypes.byref(pk) if pk is not None else None, _ptr(workspace), ws_bytes, ev,
                    int(bool(debug)), stream)
            elif split:
                dc, rest = _f32c(sh_dc), _f32c(sh_rest)
                rc = lib.g4s_rasterizer_backward_split_sh(
                    P, int(degree), M, int(R), _ptr(bg), W, H, _ptr(m3), _ptr(dc), _ptr(rest), _ptr(sc),
                    float(scale_modifier), _ptr(rot), _ptr(tm), _ptr(vm), _ptr(pm), _ptr(cp), float(tan_fovx),
                    float(tan_fovy), _ptr(rad), _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer), _ptr(gc),
                    _ptr(go), _ptr(dL_dmeans2D), _ptr(dL_dnormal), _ptr(dL_dopacity), _ptr(dL_dcolors),
                    _ptr(dL_dmeans3D), _ptr(dL_dtransMat), _ptr(dL_dsh[0]), _ptr(dL_dsh[1]), _ptr(dL_dscales),
                    _ptr(dL_drotations), _ptr(workspace), ws_bytes, int(bool(debug)), stream)
            else:
                shc = _f32c(sh, 16)
                rc = lib.g4s_rasterizer_backward(
                    P, int(degree), M, int(R), _ptr(bg), W, H, _ptr(m3), _ptr(shc), _ptr(col), _ptr(sc),
                    float(scale_modifier), _ptr(rot), _ptr(tm), _ptr(vm), _ptr(pm), _ptr(cp), float(tan_fovx),
                    float(tan_fovy), _ptr(rad), _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer), _ptr(gc),
                    _ptr(go), _ptr(dL_dmeans2D), _ptr(dL_dnormal), _ptr(dL_dopacity), _ptr(dL_dcolors),
                    _ptr(dL_dmeans3D), _ptr(dL_dtransMat), _ptr(dL_dsh), _ptr(dL_dscales), _ptr(dL_drotations),
                    _ptr(workspace), ws_bytes, int(bool(debug)), stream)
            if rc != 0:
                _raise(rc, "rasterize_gaussians_backward")
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dtransMat, dL_dsh, dL_dscales, dL_drotations


def mark_visible(means3D, viewmatrix, projmatrix):
    lib = _lib.load()
    P = int(means3D.size(0))
    present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
    if P != 0:
        _check_cuda(means3D, "means3D")
        with torch.cuda.device(means3D.device):
            m3, vm, pm = _f32c(means3D), _f32c(viewmatrix), _f32c(projmatrix)
            stream = ctypes.c_void_p(torch.cuda.current_stream(means3D.device).cuda_stream)
            rc = lib.g4s_rasterizer_mark_visible(P, _ptr(m3), _ptr(vm), _ptr(pm), _ptr(present), stream)
            if rc != 0:
                _raise(rc, "mark_visible")
    return present
