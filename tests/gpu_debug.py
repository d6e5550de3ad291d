"""Stage-by-stage HIP-vs-oracle report (not a pytest file).  `python tests/gpu_debug.py` on the GPU box."""
import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from common import cotangents, hip_state, rel_err, run_hip, run_oracle, scene_inputs
from oracle import oracle as oracle_mod

def main():
    inp = scene_inputs(P=4000, W=200, H=136, seed=11, D=3, bg=(0.1, 0.3, 0.6), scale_mul=2.0)
    g = cotangents(136, 200)
    o = run_oracle(oracle_mod, inp, g)
    h = run_hip(inp, None, debug=True)
    print("R hip/oracle", h["R"], o["R"])
    st = hip_state(h, inp); orc = o["oracle"]; vis = o["radii"] > 0
    print("radii equal", np.array_equal(h["radii"], o["radii"]), "visible", vis.sum())
    print("tiles_touched equal", np.array_equal(st["tiles_touched"], orc.state("tiles_touched")))
    rec = st["rec"]
    for nm, sl, ref in (("means2D", slice(0, 2), orc.state("means2D")), ("normal_opacity", slice(4, 8), orc.state("normal_opacity")),
                        ("transMat", slice(8, 17), orc.state("transMat")), ("rgb", slice(17, 20), orc.state("rgb"))):
        d = np.abs(rec[vis, sl] - ref[vis]); print(f"rec.{nm}: bitwise {np.array_equal(rec[vis, sl], ref[vis])} maxabs {d.max():.3e}")
    # depth order
    ds = st["depth_sorted"]; keys = np.where(vis, orc.state("depths").view(np.uint32), 0xFFFFFFFF)
    want = np.argsort(keys, kind="stable").astype(np.uint32)
    print("depth order equal", np.array_equal(ds, want))
    ent = st["entries"]
    pl = (ent & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    print("point_list equal", np.array_equal(pl, orc.state("point_list")), "tiles equal",
          np.array_equal((ent >> np.uint64(48)).astype(np.uint32), (orc.state("keys") >> np.uint64(32)).astype(np.uint32)))
    print("ranges equal", np.array_equal(st["ranges"], orc.state("ranges")))
    print("n_contrib equal", np.array_equal(st["n_contrib"], orc.state("n_contrib")), "final_T maxabs", np.abs(st["final_T"] - orc.state("final_T")).max())
    print("color maxabs", np.abs(h["color"] - o["color"]).max())
    for c in range(7): print("others", c, np.abs(h["others"][c] - o["others"][c]).max())
    h = run_hip(inp, g, debug=True)
    for k in h["grads"]:
        print("grad", k, "rel", rel_err(h["grads"][k], o["grads"][k]), "max", np.abs(o["grads"][k]).max())

if __name__ == "__main__":
    try:
        main()
    except Exception:
        traceback.print_exc(); sys.exit(1)
