"""Drop-in replacement of the `diff_surfel_rasterization` Python package on MI355X.

Public surface (names, argument order, defaults, return values and exceptions) follows
dsr/diff_surfel_rasterization/__init__.py of the reference:

    GaussianRasterizationSettings   NamedTuple, 12 fields in the reference's order   (:158-170)
    GaussianRasterizer              nn.Module with .markVisible() and .forward()     (:172-222)
    rasterize_gaussians(...)        functional form                                  (:21-42)

The autograd node calls `_C.rasterize_gaussians` / `_C.rasterize_gaussians_backward` with the
reference's argument tuples (:60-80, :109-130) and returns gradients in input order
(:144-154).  `_C` here is the ctypes front-end of the HIP library (./_C.py), not a pybind module.
"""
import contextlib
import threading
from typing import NamedTuple

import torch
from torch import nn

from . import _C

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "presized", "last_num_rendered"]

_slot = threading.local()

# num_rendered of the most recent reference-shaped forward on this thread (what the reference's forward returns and
# throws away, __init__.py:98): parallel.ViewParallel.accumulate sizes its PresizedStates from it.
_last = threading.local()


def last_num_rendered():
    """Instance count of this thread's most recent forward that read it back (None before the first one / inside slots)."""
    return getattr(_last, "value", None)


@contextlib.contextmanager
def presized(state, workspace=None):
    """Extension over the reference (g4splat_amd.pipeline.ViewPipeline.slot enters it): inside this context the autograd
    node renders through `_C.rasterize_gaussians_presized` with `state` (a `_C.PresizedState`: the forward's scratch lives
    in it and the host never waits for `num_rendered`) and its backward uses `workspace` (uint8 tensor of at least
    g4s_rasterizer_backward_workspace(P, state.capacity) bytes) instead of allocating one.  The node's forward state IS
    `state`: run this view's backward before the next forward that uses the same state, and check `state.status[3]`
    (instance capacity exceeded: that frame is invalid) when convenient."""
    prev = getattr(_slot, "value", None)
    _slot.value = (state, workspace)
    try:
        yield
    finally:
        _slot.value = prev


def _forward(*fwd_args):
    """(outputs of _C.rasterize_gaussians, backward context or None) through the presized entry when a slot is active.
    The backward context is (workspace, state, the state's generation after this forward)."""
    slot = getattr(_slot, "value", None)
    if slot is None:
        out = _C.rasterize_gaussians(*fwd_args)
        _last.value = int(out[0])
        return out, None
    state = slot[0]
    out = _C.rasterize_gaussians_presized(state, *fwd_args)
    state.generation = getattr(state, "generation", 0) + 1
    return out, (slot[1], state, state.generation)


def _backward(bctx):
    def call(*bwd_args):
        # (cov3Ds_precomp / transMat not given: its gradient is an intermediate of the native backward that nobody reads)
        out = {"dL_dtransMat": False} if bwd_args[7].numel() == 0 else {}
        if bctx is None:
            return _C.rasterize_gaussians_backward(*bwd_args, out=out or None)
        workspace, state, generation = bctx
        if getattr(state, "generation", generation) != generation:
            # the forward's scratch IS the PresizedState: a later forward through the same state has overwritten it
            raise RuntimeError("backward of a view whose PresizedState has been used by another forward since "
                               "(run a view's backward before its slot renders the next view)")
        if workspace is not None:
            out["workspace"] = workspace
        return _C.rasterize_gaussians_backward(*bwd_args, out=out or None)
    return call


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _snapshot(args):
    """Host copy of an argument tuple, taken before a debug-mode launch can corrupt it (:17-19)."""
    def host(a):
        if isinstance(a, torch.Tensor):
            return a.detach().cpu().clone()
        if isinstance(a, (tuple, list)):  # the split-SH pair (features_dc, features_rest)
            return tuple(host(x) for x in a)
        return a
    return tuple(host(a) for a in args)


def _call_guarded(fn, args, debug, dump_name, banner):
    """debug=True: on failure dump the CPU snapshot of the arguments and re-raise (:83-90, :133-140)."""
    if not debug:
        return fn(*args)
    saved = _snapshot(args)
    try:
        return fn(*args)
    except Exception:
        torch.save(saved, dump_name)
        print(banner)
        raise


def _present(grad_color, grad_depth, rs, device):
    """An output the loss did not use has no cotangent (None): the native side wants both maps."""
    if grad_color is None:
        grad_color = torch.zeros((3, rs.image_height, rs.image_width), dtype=torch.float32, device=device)
    if grad_depth is None:
        grad_depth = torch.zeros((7, rs.image_height, rs.image_width), dtype=torch.float32, device=device)
    return grad_color, grad_depth


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        rs = raster_settings
        fwd_args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                    rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
                    rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
        (num_rendered, color, depth, radii, geomBuffer, binningBuffer, imgBuffer), ctx.workspace = _call_guarded(
            _forward, fwd_args, rs.debug, "snapshot_fw.dump",
            "\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_depth):
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer,
         imgBuffer) = ctx.saved_tensors
        grad_out_color, grad_depth = _present(grad_out_color, grad_depth, rs, means3D.device)
        bwd_args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                    rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, grad_depth, sh,
                    rs.sh_degree, rs.campos, geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, rs.debug)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = _call_guarded(
            _backward(ctx.workspace), bwd_args, rs.debug, "snapshot_bw.dump",
            "\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
        # one gradient per forward input, in input order; raster_settings gets None
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales,
                grad_rotations, grad_cov3Ds_precomp, None)


class _RasterizeGaussiansSplitSH(torch.autograd.Function):
    """Same node with the SH coefficients given as the model's two parameter tensors (features_dc [P,1,3],
    features_rest [P,M-1,3]) instead of their concatenation: the library reads them in place and writes the two
    gradients separately (include/g4s_rasterizer.h, g4s_rasterizer_*_split_sh).  Extension over the reference."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh_dc, sh_rest, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        rs = raster_settings
        empty = torch.empty(0, dtype=torch.float32, device=means3D.device)
        fwd_args = (rs.bg, means3D, empty, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                    rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width,
                    (sh_dc, sh_rest), rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
        (num_rendered, color, depth, radii, geomBuffer, binningBuffer, imgBuffer), ctx.workspace = _call_guarded(
            _forward, fwd_args, rs.debug, "snapshot_fw.dump",
            "\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(means3D, scales, rotations, cov3Ds_precomp, radii, sh_dc, sh_rest, geomBuffer,
                              binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_depth):
        rs = ctx.raster_settings
        (means3D, scales, rotations, cov3Ds_precomp, radii, sh_dc, sh_rest, geomBuffer, binningBuffer,
         imgBuffer) = ctx.saved_tensors
        grad_out_color, grad_depth = _present(grad_out_color, grad_depth, rs, means3D.device)
        empty = torch.empty(0, dtype=torch.float32, device=means3D.device)
        bwd_args = (rs.bg, means3D, radii, empty, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
                    rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, grad_depth, (sh_dc, sh_rest), rs.sh_degree,
                    rs.campos, geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, rs.debug)
        (grad_means2D, _grad_colors, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = _call_guarded(
            _backward(ctx.workspace), bwd_args, rs.debug, "snapshot_bw.dump",
            "\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
        return (grad_means3D, grad_means2D, grad_sh[0], grad_sh[1], grad_opacities, grad_scales, grad_rotations,
                grad_cov3Ds_precomp, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    if isinstance(sh, (tuple, list)):
        if colors_precomp is not None and colors_precomp.numel():
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        return _RasterizeGaussiansSplitSH.apply(means3D, means2D, sh[0], sh[1], opacities, scales, rotations,
                                                cov3Ds_precomp, raster_settings)
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """bool[P]: Gaussians in front of the near plane of this camera (no gradient)."""
        rs = self.raster_settings
        with torch.no_grad():
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        has_sr = scales is not None or rotations is not None
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (has_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

        def absent():  # empty tensor == "not given" for the native side
            return torch.empty(0, dtype=torch.float32, device=means3D.device)

        shs = absent() if shs is None else shs  # (features_dc, features_rest) is passed through: split-SH extension
        colors_precomp = absent() if colors_precomp is None else colors_precomp
        scales = absent() if scales is None else scales
        rotations = absent() if rotations is None else rotations
        cov3D_precomp = absent() if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   rs)
