import torch, faulthandler
faulthandler.enable()
dev = torch.device("cuda", 0)
a = torch.ones(1 << 20, device=dev); b = torch.ones(1 << 20, device=dev)
side = torch.cuda.Stream(device=dev)
# warm-up
s0 = torch.cuda.Stream(device=dev); s0.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s0):
    side.wait_stream(s0)
    with torch.cuda.stream(side):
        c = a * 2
    s0.wait_stream(side)
    d = c + b
torch.cuda.current_stream().wait_stream(s0); torch.cuda.synchronize()
for variant in ("wait_stream", "event"):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        if variant == "wait_stream":
            side.wait_stream(cur)
        else:
            e0 = torch.cuda.Event(); e0.record(cur); side.wait_event(e0)
        with torch.cuda.stream(side):
            c = a * 2
            e1 = torch.cuda.Event(); e1.record(side)
        if variant == "wait_stream":
            cur.wait_stream(side)
        else:
            cur.wait_event(e1)
        d = c + b
    g.replay(); torch.cuda.synchronize()
    print(variant, "ok", float(d.sum()))
