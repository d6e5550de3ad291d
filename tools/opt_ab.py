import sys, json, io, contextlib
sys.path.insert(0, "/root/repo")
from g4splat_amd import _lib
import bench
for rep in range(2):
    for v in (0, 1):
        _lib.set_option("bwd_fwd_order", v)
        sys.argv = ["bench.py", "--steps", "60", "--warmup", "10", "--no-cpu-baseline", "--sustained-seconds", "0", "--views-in-flight", "0"]
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.main()
        d = [json.loads(l) for l in buf.getvalue().splitlines() if l.startswith("{")][0]
        print("bwd_fwd_order", v, "ms/step %.4f" % d["ms_per_step"], "blend_bwd", d["kernels_ms"]["blend_bwd"], flush=True)
