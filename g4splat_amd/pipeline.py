"""Several independent views in flight on one GPU (new functionality; the reference renders one view per optimiser step).

The views of one multi-view batch -- gradient accumulation, or SURVEY.md 8(e)'s 8 training views over fewer than 8 GPUs --
do not depend on each other.  One view's step is a chain of launch-bound binning kernels, two VALU-bound blend kernels and
HBM-bound per-Gaussian kernels; put K of those chains on K HIP streams and the GPU runs one view's binning and
per-Gaussian kernels beside another view's blend kernels.  Measured on MI355X: S3 1.99 -> 1.76 ms per
view at K = 3, S2 0.59 -> 0.37, S5 2.75 -> 1.92; every view's results are bit-identical whatever runs beside it
(tests/test_gpu_dp.py).

    pipe = ViewPipeline(P, W, H, instance_capacity, device, k=2)
    for i, view in enumerate(batch):
        with pipe.slot(i) as (state, workspace):          # stream i % k is current inside the block
            fw = _C.rasterize_gaussians_presized(state, ...view...)
            grads[i] = _C.rasterize_gaussians_backward(..., fw[4], fw[0], fw[5], fw[6], False, out={"workspace": workspace, ...})
    pipe.join()                                           # the caller's stream waits for all of them

Through the autograd node (the reference's `render()` path) the slot is picked up by itself:

    pipe.order_accumulation(gaussians.parameters())      # once: see below
    for i, view in enumerate(batch):
        with pipe.slot(i):                                # diff_surfel_rasterization.presized(state, workspace) is active
            out = render(view, gaussians, pipe_cfg, bg)   # gaussian_renderer.render, unchanged
            loss_of(out, view).backward()                 # runs on the slot's stream (autograd replays forward streams)
            pipe.after_previous_view()                    # before touching anything the views share:
            gaussians.add_densification_stats(out["viewspace_points"], out["visibility_filter"], out["radii"])
    pipe.join(); optimizer.step()

What the views SHARE must be ordered by hand, because two backward() calls on two streams are two unrelated graphs to
autograd: each accumulates into the leaves' .grad on its own stream, and `grad += g` from two streams at once loses
updates (seen as run-to-run differences of the parameters, not as an error).  `order_accumulation(params)` registers a hook
on every leaf that makes the accumulating stream wait for the previous view's slot to finish (an event recorded when a slot
exits; by then that view's backward has been issued), so the accumulations happen one after the other in program order and
the sums are bit-identical to the sequential loop's.  `after_previous_view()` is the same wait for user code (densification
statistics, running means).  The waits sit at the very end of a view's backward: nothing of the overlap is lost.
(The leaves' .grad buffers must be persistent -- `order_accumulation` allocates missing ones on the caller's stream and
`slot()` refuses to start a view when one has been dropped (zero_grad(set_to_none=True)): a gradient born inside a slot
would move from that stream's allocator pool to the optimiser reading it on another stream; a view's backward must be issued before its slot is used again -- the
autograd node checks that and raises otherwise.  Anything else produced inside a slot and read on another stream after
join(), an image kept for logging say, follows torch's rule for cross-stream use: `t.record_stream(reader_stream)`.)

Every slot owns a PresizedState (g4s_rasterizer_forward_presized: no host read-back, so the one host thread never blocks),
a backward workspace and -- through torch's per-stream allocator pools -- its output tensors.  The backward OVERWRITES its
outputs: gradients of different views must land in different buffers and be summed afterwards."""
import contextlib

import torch

from . import _lib
from .diff_surfel_rasterization import _C


class ViewPipeline:
    def __init__(self, P, width, height, instance_capacity, device, k=2):
        if k < 1:
            raise ValueError("k must be >= 1")
        self.device = torch.device(device)
        self.k = int(k)
        lib = _lib.load()
        ws_bytes = lib.g4s_rasterizer_backward_workspace(int(P), int(instance_capacity))
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.k)]
        self.states, self.workspaces = [], []
        self._done = [torch.cuda.Event() for _ in range(self.k)]  # recorded when a slot exits
        self._last_done = None
        self._hooks = []
        self._ordered = []  # the leaves of order_accumulation: their .grad buffers must stay persistent
        # per slot: the element-wise maximum of the status words [num_rendered, binned, emitting, overflow] of every forward
        # issued in the slot since reset_peak() -- a slot serves several views of a step, and its state's own status word
        # only remembers the last one
        self._peak = []
        for s in self.streams:
            with torch.cuda.stream(s):
                self.states.append(_C.PresizedState(P, width, height, instance_capacity, self.device))
                self.workspaces.append(torch.empty(ws_bytes, dtype=torch.uint8, device=self.device))
                self._peak.append(torch.zeros(4, dtype=torch.int32, device=self.device))
        torch.cuda.synchronize(self.device)

    @contextlib.contextmanager
    def slot(self, i):
        """Makes stream i % k current (after it has waited for the caller's stream: parameters written there are visible)
        and yields that slot's (PresizedState, backward workspace)."""
        j = i % self.k
        s = self.streams[j]
        for p in self._ordered:
            if p.grad is None:
                # a .grad created inside a slot would come from the slot stream's allocator pool and be read by the
                # optimiser on the caller's stream: freed back to that pool by zero_grad(set_to_none=True), it could be
                # handed out again while the optimiser is still reading it
                raise RuntimeError("ViewPipeline: a parameter registered with order_accumulation() has no .grad buffer "
                                   "(zero_grad(set_to_none=True)?): keep the gradients persistent -- "
                                   "optimizer.zero_grad(set_to_none=False) or GradientBucket.zero()")
        s.wait_stream(torch.cuda.current_stream(self.device))
        from .diff_surfel_rasterization import presized
        with torch.cuda.stream(s), presized(self.states[j], self.workspaces[j]):
            try:
                yield self.states[j], self.workspaces[j]
            finally:
                torch.maximum(self._peak[j], self.states[j].status, out=self._peak[j])  # (on the slot's stream)
                self._done[j].record(s)
                self._last_done = self._done[j]

    @property
    def previous_view_done(self):
        """The event recorded when the most recently exited slot was left (None before the first one): what
        `out["after"]` of an accumulating `rasterize_gaussians_backward` takes -- the accumulating kernel of this view then
        waits for the previous view's, and nothing else of this view does."""
        return self._last_done

    def after_previous_view(self):
        """The current stream waits until everything issued inside the most recently exited slot has finished.  Call it
        inside a slot before updating state that the views share."""
        ev = self._last_done
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)

    def order_accumulation(self, params):
        """Registers a gradient hook on every leaf in `params`: its accumulation into .grad waits for the previous view
        (see the module docstring).  Returns the hook handles (also kept in the pipeline; `release_hooks()` removes them)."""
        def hook(grad):
            self.after_previous_view()
            return grad
        params = list(params)
        for p in params:
            if p.grad is None:  # persistent buffers from the caller's stream, so that no gradient is born in a slot's pool
                p.grad = torch.zeros_like(p)
        handles = [p.register_hook(hook) for p in params]
        self._hooks.extend(handles)
        self._ordered.extend(params)
        return handles

    def release_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self._ordered = []

    def join(self):
        """The caller's current stream waits for every slot's stream."""
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            cur.wait_stream(s)

    def reset_peak(self):
        """Forgets the status words seen so far (start of a step)."""
        for j, s in enumerate(self.streams):
            with torch.cuda.stream(s):
                self._peak[j].zero_()

    def peak(self):
        """(largest num_rendered, overflowed) over every forward issued in a slot since reset_peak() -- not only each
        slot's last one.  Reads the device words: synchronises with the slots' streams."""
        words = torch.stack(self._peak).cpu()
        return int(words[:, 0].max()), bool(words[:, 3].max() != 0)

    def overflowed(self):
        """True if a forward issued in some slot since reset_peak() exceeded the instance capacity (synchronises)."""
        return self.peak()[1]
