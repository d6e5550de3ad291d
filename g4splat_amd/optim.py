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
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        # (betas, eps, device) -> segments; the reference uses one setting for all groups => one launch per 8 tensors
        batches = {}
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam: parameters must be CUDA tensors (use torch.optim.Adam on the host)")
                if p.grad.is_sparse or p.dtype != torch.float32:
                    raise RuntimeError("FusedAdam supports dense float32 parameters only")
                st = self.state[p]
                if len(st) == 0:  # same lazy state initialisation as torch.optim.Adam
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                for name in ("exp_avg", "exp_avg_sq"):
                    if not st[name].is_contiguous():
                        st[name] = st[name].contiguous()
                if not p.is_contiguous():
                    raise RuntimeError("FusedAdam: parameters must be contiguous")
                key = (group["betas"][0], group["betas"][1], group["eps"], p.device)
                batches.setdefault(key, []).append((p, p.grad.contiguous(), st, float(group["lr"])))
        for (b1, b2, eps, dev), items in batches.items():
            with torch.cuda.device(dev):
                stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                for i in range(0, len(items), 8):
                    seg = items[i:i + 8]
                    k = len(seg)
                    ptr = lambda ts: (ctypes.c_void_p * k)(*[t.data_ptr() for t in ts])
                    rc = lib.g4s_adam_step(
                        k, ptr([s[0] for s in seg]), ptr([s[1] for s in seg]), ptr([s[2]["exp_avg"] for s in seg]),
                        ptr([s[2]["exp_avg_sq"] for s in seg]), (ctypes.c_longlong * k)(*[s[0].numel() for s in seg]),
                        (ctypes.c_double * k)(*[s[3] for s in seg]), (ctypes.c_int * k)(*[int(s[2]["step"]) for s in seg]),
                        float(b1), float(b2), float(eps), stream)
                    if rc != 0:
                        raise RuntimeError(f"g4s_adam_step failed ({rc}): {_lib.last_error()}")
        return loss
