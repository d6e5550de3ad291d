"""Fused Adam for the Gaussian parameter groups (include/g4s_optim.h).

`FusedAdam` is a `torch.optim.Optimizer` with torch.optim.Adam's hyper-parameters, param_groups and per-parameter
state layout (`step`, `exp_avg`, `exp_avg_sq`), so the reference's densification code -- which edits
`optimizer.state[p]["exp_avg"]` directly (2dgs/scene/gaussian_model.py:495-560) -- keeps working; only `step()`
differs: all groups are updated by ONE HIP kernel instead of torch's foreach passes.  The reference builds its
optimiser at gaussian_model.py:248-266 (six groups, eps = 1e-15); GaussianModel.training_setup(fused=True) does the
same with this class.  HIP tensors only (no CPU path): use torch.optim.Adam for host tensors."""
import ctypes

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    """capturable=True (extension, same meaning as torch.optim.Adam's): state["step"] is a float32 scalar ON THE DEVICE and
    the learning rates are read from a device buffer, so that a step() captured in a HIP graph (graphed.TrainStepGraph)
    does the right update at every replay.  Outside a capture step() uploads the groups' current `lr` itself; a replayed
    graph cannot, so call `sync_lr()` before `graph.replay()` whenever a learning rate has changed."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, *, foreach=None,
                 maximize=False, capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError("invalid Adam hyper-parameter")
        # torch.optim.Adam's full set of param-group keys, so that state_dict()s move between the two classes in both
        # directions (GaussianModel.capture() / restore(), and the reference's own restore()); the kernel implements
        # the plain update only, so anything but the defaults is refused -- here and again in step(), because
        # load_state_dict() overwrites the groups.
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, maximize=maximize,
                        foreach=foreach, capturable=capturable, differentiable=differentiable, fused=fused,
                        decoupled_weight_decay=decoupled_weight_decay)
        self._check_plain(defaults)
        super().__init__(params, defaults)
        self._dev = {}  # capturable: (batch key, chunk) -> (lr_dev, last uploaded values, coef_dev)

    @staticmethod
    def _check_plain(group):
        if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False) or \
                group.get("differentiable", False) or group.get("decoupled_weight_decay", False):
            raise ValueError("FusedAdam implements plain Adam only: weight_decay, amsgrad, maximize, differentiable and "
                             "decoupled_weight_decay must keep their defaults")

    def _batches(self, advance_host_step, need_grad=True):
        """(betas, eps, device, capturable) -> [(param, grad or None, state, lr)], state created lazily like
        torch.optim.Adam's.  need_grad: leave parameters without a gradient out (the plain path); the capturable path keeps
        them as empty segments so that a parameter's slot in the device-side learning-rate buffer never moves."""
        batches = {}
        for group in self.param_groups:
            self._check_plain(group)
            cap = bool(group.get("capturable", False))
            for p in group["params"]:
                key = (group["betas"][0], group["betas"][1], group["eps"], p.device, cap)
                if p.grad is None:
                    if cap and not need_grad and p.requires_grad:
                        batches.setdefault(key, []).append((p, None, self.state[p], float(group["lr"])))
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam: parameters must be CUDA tensors (use torch.optim.Adam on the host)")
                if p.grad.is_sparse or p.dtype != torch.float32:
                    raise RuntimeError("FusedAdam supports dense float32 parameters only")
                st = self.state[p]
                if len(st) == 0:  # same lazy state initialisation as torch.optim.Adam
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device) if cap else torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if cap and not st["step"].is_cuda:  # (a state dict loaded from a non-capturable optimiser)
                    st["step"] = st["step"].to(device=p.device, dtype=torch.float32)
                if advance_host_step and not cap:
                    st["step"] += 1
                for name in ("exp_avg", "exp_avg_sq"):
                    if not st[name].is_contiguous():
                        st[name] = st[name].contiguous()
                if not p.is_contiguous():
                    raise RuntimeError("FusedAdam: parameters must be contiguous")
                batches.setdefault(key, []).append((p, p.grad.contiguous(), st, float(group["lr"])))
        return batches

    def _dev_buffers(self, key, n, dev):
        """capturable: per batch key, one learning rate per parameter slot (slot = position among the key's trainable
        parameters), the values last uploaded, 16 floats of bias-correction scratch per chunk of eight and a dummy step."""
        buf = self._dev.get(key)
        if buf is None or buf[0].numel() != n:
            chunks = (n + 7) // 8
            buf = (torch.zeros(n, dtype=torch.float64, device=dev), [None] * n,
                   torch.zeros(16 * chunks, dtype=torch.float32, device=dev), torch.zeros((), dtype=torch.float32, device=dev))
            self._dev[key] = buf
        return buf

    @torch.no_grad()
    def sync_lr(self):
        """capturable: uploads the groups' current learning rates to the device buffer the (captured) step reads.  On the
        current stream; call it before replaying a graph that contains step()."""
        for key, items in self._batches(False, need_grad=False).items():
            if not key[4]:
                continue
            with torch.cuda.device(key[3]):
                lr_dev, last, _coef, _dummy = self._dev_buffers(key, len(items), key[3])
                new = [s[3] for s in items]
                if new != last:
                    # from PAGEABLE memory: the runtime stages the floats before the call returns, so the host may change
                    # its copy while the GPU is still iterations behind (a reused pinned buffer would race)
                    lr_dev.copy_(torch.tensor(new, dtype=torch.float64))
                    last[:] = new

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        # (betas, eps, device) -> segments; the reference uses one setting for all groups => one launch per 8 tensors
        batches = self._batches(True, need_grad=False)
        if any(k[4] for k in batches) and not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        for (b1, b2, eps, dev, cap), items in batches.items():
            with torch.cuda.device(dev):
                stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                if cap:
                    lr_dev, _last, coef, dummy = self._dev_buffers((b1, b2, eps, dev, cap), len(items), dev)
                for c, i in enumerate(range(0, len(items), 8)):
                    seg = items[i:i + 8]
                    k = len(seg)
                    ptr = lambda ts: (ctypes.c_void_p * k)(*[(t.data_ptr() if t is not None else 0) for t in ts])
                    if cap:
                        if all(s[1] is None for s in seg):
                            continue
                        live = [s[1] is not None for s in seg]
                        rc = lib.g4s_adam_step_device(
                            k, ptr([s[0] for s in seg]), ptr([s[1] for s in seg]),
                            ptr([s[2]["exp_avg"] if ok else None for s, ok in zip(seg, live)]),
                            ptr([s[2]["exp_avg_sq"] if ok else None for s, ok in zip(seg, live)]),
                            (ctypes.c_longlong * k)(*[(s[0].numel() if ok else 0) for s, ok in zip(seg, live)]),
                            ctypes.c_void_p(lr_dev.data_ptr() + 8 * i),
                            ptr([s[2]["step"] if ok else dummy for s, ok in zip(seg, live)]),  # (no gradient: t stays)
                            ctypes.c_void_p(coef.data_ptr() + 64 * c), float(b1), float(b2), float(eps), stream)
                    else:
                        rc = lib.g4s_adam_step(
                            k, ptr([s[0] for s in seg]), ptr([s[1] for s in seg]), ptr([s[2]["exp_avg"] for s in seg]),
                            ptr([s[2]["exp_avg_sq"] for s in seg]), (ctypes.c_longlong * k)(*[s[0].numel() for s in seg]),
                            (ctypes.c_double * k)(*[s[3] for s in seg]), (ctypes.c_int * k)(*[int(s[2]["step"]) for s in seg]),
                            float(b1), float(b2), float(eps), stream)
                    if rc != 0:
                        raise RuntimeError(f"g4s_adam_step failed ({rc}): {_lib.last_error()}")
        return loss


@torch.no_grad()
def densify_stats(grad_mean2D, update_filter, xyz_gradient_accum, denom, radii=None, max_radii2D=None):
    """One-kernel form of GaussianModel.add_densification_stats (gaussian_model.py:649-651) plus, when `radii` and
    `max_radii2D` are given, the training loop's max_radii2D update (include/g4s_optim.h, g4s_densify_stats).
    All tensors on one HIP device; statistics are updated in place."""
    P = int(grad_mean2D.shape[0])
    dev = grad_mean2D.device
    if not grad_mean2D.is_cuda:
        raise RuntimeError("densify_stats: HIP tensors only")
    if (tuple(grad_mean2D.shape) != (P, 3) or grad_mean2D.dtype != torch.float32 or update_filter.dtype != torch.bool
            or update_filter.numel() != P or xyz_gradient_accum.numel() != P or denom.numel() != P):
        raise RuntimeError("densify_stats: expected grad [P,3] float32, filter bool [P], statistics float32 [P(,1)]")
    for t in (xyz_gradient_accum, denom) + ((max_radii2D,) if max_radii2D is not None else ()):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
            raise RuntimeError("densify_stats: statistics must be contiguous float32 tensors on the gradient's device")
    if (radii is None) != (max_radii2D is None):
        raise RuntimeError("densify_stats: give radii and max_radii2D together")
    g, f = grad_mean2D.contiguous(), update_filter.contiguous()
    r = radii.contiguous() if radii is not None else None
    if r is not None and (r.dtype != torch.int32 or r.numel() != P or max_radii2D.numel() != P):
        raise RuntimeError("densify_stats: radii must be int32 [P]")
    lib = _lib.load()
    with torch.cuda.device(dev):
        rc = lib.g4s_densify_stats(P, ctypes.c_void_p(g.data_ptr()), ctypes.c_void_p(f.data_ptr()),
                                   ctypes.c_void_p(r.data_ptr() if r is not None else 0),
                                   ctypes.c_void_p(xyz_gradient_accum.data_ptr()), ctypes.c_void_p(denom.data_ptr()),
                                   ctypes.c_void_p(max_radii2D.data_ptr() if max_radii2D is not None else 0),
                                   ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"g4s_densify_stats failed ({rc}): {_lib.last_error()}")


class _Activations(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scaling, rotation, opacity):
        lib = _lib.load()
        dev = scaling.device
        P = int(scaling.shape[0])
        s, r, o = scaling.detach().contiguous(), rotation.detach().contiguous(), opacity.detach().contiguous()
        with torch.cuda.device(dev):
            scales, rots, opac = torch.empty_like(s), torch.empty_like(r), torch.empty_like(o)
            rc = lib.g4s_activations_forward(P, ctypes.c_void_p(s.data_ptr()), ctypes.c_void_p(r.data_ptr()),
                                             ctypes.c_void_p(o.data_ptr()), ctypes.c_void_p(scales.data_ptr()),
                                             ctypes.c_void_p(rots.data_ptr()), ctypes.c_void_p(opac.data_ptr()),
                                             ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"g4s_activations_forward failed ({rc}): {_lib.last_error()}")
        ctx.save_for_backward(scales, r, opac)
        return scales, rots, opac

    @staticmethod
    def backward(ctx, g_scales, g_rots, g_opac):
        scales, r, opac = ctx.saved_tensors
        lib = _lib.load()
        dev = scales.device
        P = int(scales.shape[0])
        zero = lambda ref, g: torch.zeros_like(ref) if g is None else g.contiguous()
        gs, gr, go = zero(scales, g_scales), zero(r, g_rots), zero(opac, g_opac)
        with torch.cuda.device(dev):
            ds, dr, do = torch.empty_like(scales), torch.empty_like(r), torch.empty_like(opac)
            rc = lib.g4s_activations_backward(P, ctypes.c_void_p(scales.data_ptr()), ctypes.c_void_p(r.data_ptr()),
                                              ctypes.c_void_p(opac.data_ptr()), ctypes.c_void_p(gs.data_ptr()),
                                              ctypes.c_void_p(gr.data_ptr()), ctypes.c_void_p(go.data_ptr()),
                                              ctypes.c_void_p(ds.data_ptr()), ctypes.c_void_p(dr.data_ptr()),
                                              ctypes.c_void_p(do.data_ptr()),
                                              ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"g4s_activations_backward failed ({rc}): {_lib.last_error()}")
        return ds, dr, do


def fused_activations(scaling, rotation, opacity):
    """-> (exp(scaling) [P,2], normalize(rotation) [P,4], sigmoid(opacity) [P,1]): the three getters render() reads
    (gaussian_model.py:157-192, mip filter off) as one HIP launch each way (include/g4s_optim.h).  HIP tensors only."""
    if not (scaling.is_cuda and rotation.is_cuda and opacity.is_cuda):
        raise RuntimeError("fused_activations: HIP tensors only")
    P = scaling.shape[0]
    if (tuple(scaling.shape) != (P, 2) or tuple(rotation.shape) != (P, 4) or opacity.numel() != P
            or any(t.dtype != torch.float32 for t in (scaling, rotation, opacity))):
        raise RuntimeError("fused_activations: expected float32 scaling [P,2], rotation [P,4], opacity [P,1]")
    return _Activations.apply(scaling, rotation, opacity)


@torch.no_grad()
def compact_rows(keep, tensors, extra_rows=0):
    """Stream compaction on the HIP device (include/g4s_optim.h, g4s_compact_scan / g4s_compact_gather): the rows
    `keep` marks (bool [P]) of every tensor in `tensors` (float32, contiguous, [P, ...]), in index order -- what
    `t[keep]` returns for each of them, with ONE scan of the mask and one gather launch per eight tensors instead of a
    mask -> index conversion and an indexing kernel per tensor.  Returns (n_kept, new tensors); each new tensor has
    n_kept + extra_rows rows (the tail is uninitialised: the caller appends there).  One host read-back (n_kept: the
    new tensors' shapes need it), like the reference's mask indexing."""
    P = int(keep.shape[0])
    dev = keep.device
    if not keep.is_cuda:
        raise RuntimeError("compact_rows: HIP tensors only")
    if keep.dtype != torch.bool or keep.ndim != 1:
        raise RuntimeError("compact_rows: keep must be bool [P]")
    src = []
    for t in tensors:
        if t.dtype != torch.float32 or t.shape[0] != P or t.device != dev:
            raise RuntimeError("compact_rows: tensors must be float32 [P, ...] on the mask's device")
        src.append(t.detach().contiguous())
    lib = _lib.load()
    keep_c = keep.contiguous()
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        nws = lib.g4s_compact_workspace(P)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        count = torch.zeros(1, dtype=torch.int32, device=dev)
        rc = lib.g4s_compact_scan(P, ctypes.c_void_p(keep_c.data_ptr()), ctypes.c_void_p(count.data_ptr()),
                                  ctypes.c_void_p(ws.data_ptr()), nws, stream)
        if rc != 0:
            raise RuntimeError(f"g4s_compact_scan failed ({rc}): {_lib.last_error()}")
        n = int(count.item())
        out = [torch.empty((n + int(extra_rows),) + tuple(t.shape[1:]), dtype=torch.float32, device=dev) for t in src]
        k = len(src)
        if k and P and n > 0:  # (n == 0: nothing to copy, and empty outputs have no address)
            widths = [int(t[0].numel()) if t.ndim > 1 else 1 for t in src]
            rc = lib.g4s_compact_gather(P, ctypes.c_void_p(keep_c.data_ptr()), ctypes.c_void_p(ws.data_ptr()), k,
                                        (ctypes.c_void_p * k)(*[t.data_ptr() for t in src]),
                                        (ctypes.c_void_p * k)(*[t.data_ptr() for t in out]), (ctypes.c_int * k)(*widths), 0,
                                        stream)
            if rc != 0:
                raise RuntimeError(f"g4s_compact_gather failed ({rc}): {_lib.last_error()}")
    return n, out
