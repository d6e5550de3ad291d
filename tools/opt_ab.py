"""A/B of a library option on one workload (run on the GPU box):
    python tools/opt_ab.py <option> <workload> <value> [<value> ...]      e.g.  bwd_hot_threshold s2 unset 150 250 400
Prints the headline ms/step and the blend backward's time for every value, twice."""
import contextlib
import io
import json
import sys

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from g4splat_amd import _lib  # noqa: E402
import bench  # noqa: E402

name, workload, values = sys.argv[1], sys.argv[2], sys.argv[3:]
for rep in range(2):
    for v in values:
        _lib.set_option(name, _lib.OPTION_UNSET if v == "unset" else int(v))
        sys.argv = ["bench.py", "--workload", workload, "--steps", "60", "--warmup", "10", "--no-cpu-baseline", "--sustained-seconds", "0",
                    "--views-in-flight", "0"]
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.main()
        d = [json.loads(ln) for ln in buf.getvalue().splitlines() if ln.startswith("{")][0]
        print(name, v, "ms/step %.4f" % d["ms_per_step"], "blend_bwd", d["kernels_ms"]["blend_bwd"], flush=True)
_lib.set_option(name, _lib.OPTION_UNSET)
