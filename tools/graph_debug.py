import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from types import SimpleNamespace
import test_gpu_train as T
from g4splat_amd.gaussian_renderer import render
from g4splat_amd.losses import photometric_loss, geometry_regularizers
from g4splat_amd.graphed import TrainStepGraph
from g4splat_amd.diff_surfel_rasterization import _C, presized
from g4splat_amd import _lib
dev = torch.device("cuda", 0)
cams = T._cams(dev)[:3]
g = torch.Generator(device=dev).manual_seed(3)
gts = [torch.rand((3, T.H, T.W), device=dev, generator=g) for _ in cams]
def body(out, gt):
    loss, _l1, _s = photometric_loss(out["render"], gt, 0.2)
    nm, dm = geometry_regularizers(out["rend_normal"], out["surf_normal"], out["rend_dist"])
    return loss + 0.05 * nm + 100.0 * dm
pipe = SimpleNamespace(depth_ratio=0.0, compute_cov3D_python=False); bg = torch.zeros(3, device=dev)
def fresh():
    m = T._model(0, dev, jitter=True); m.training_setup(capturable=True); return m
def eager(m, ctx, n=1, keepgrad=False):
    grads = None
    for it in range(n):
        m.update_learning_rate(it + 1)
        with ctx():
            out = render(cams[it % 3], m, pipe, bg); loss = body(out, gts[it % 3]); loss.backward()
        grads = [p.grad.detach().clone() for p in m.parameters()]
        m.optimizer.step(); m.optimizer.zero_grad(set_to_none=True)
    return grads
import contextlib
a = fresh(); ga = eager(a, contextlib.nullcontext)
b = fresh()
lib = _lib.load()
st = _C.PresizedState(T.P, T.W, T.H, 200000, dev); ws = torch.empty(lib.g4s_rasterizer_backward_workspace(T.P, 200000), dtype=torch.uint8, device=dev)
gb = eager(b, lambda: presized(st, ws))
names = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
for n, x, y in zip(names, ga, gb): print("grad regular vs presized", n, torch.equal(x, y), float((x - y).abs().max()))
for n, x, y in zip(names, a.parameters(), b.parameters()): print("param regular vs presized", n, torch.equal(x, y), float((x - y).abs().max()))
c = fresh()
step = TrainStepGraph(c, body, cams[0], (3, T.H, T.W), instance_capacity=200000)
c.update_learning_rate(1); step(cams[0], gts[0]); torch.cuda.synchronize()
for n, x, y in zip(names, b.parameters(), c.parameters()): print("param presized-eager vs graph", n, torch.equal(x, y), float((x - y).abs().max()))
for n, pb, pc in zip(names, b.parameters(), c.parameters()):
    sb, sc = b.optimizer.state[pb], c.optimizer.state[pc]
    print(" state", n, float(sb["step"]), float(sc["step"]), torch.equal(sb["exp_avg"], sc["exp_avg"]), torch.equal(sb["exp_avg_sq"], sc["exp_avg_sq"]))
for name, m in (("eager", b), ("graph", c)):
    for k, v in m.optimizer._dev.items():
        print(name, "lr_dev", [repr(float(x)) for x in v[0].cpu()], "last", v[1], "coef", [repr(float(x)) for x in v[2].cpu()[:6]])
    print(name, "group lr", [repr(gr["lr"]) for gr in m.optimizer.param_groups])
