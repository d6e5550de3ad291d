"""Shared helpers of the parity tests: seeded inputs, oracle / HIP drivers, error metrics."""
import numpy as np

from g4splat_amd import synthetic

EMPTY = np.zeros((0,), np.float32)


def scene_inputs(P=10000, W=256, H=256, seed=0, D=3, bg=(0.0, 0.0, 0.0), scale_mul=1.0, opacity_max=1.0,
                 scale_modifier=1.0, fov_deg=60.0):
    scene, cam = synthetic.scene_random(P, seed=seed, width=W, height=H, opacity_max=opacity_max, fov_deg=fov_deg)
    return dict(bg=np.asarray(bg, np.float32), means3D=scene.means3D, colors=EMPTY, opacity=scene.opacities,
                scales=(scene.scales * scale_mul).astype(np.float32), rotations=scene.rotations,
                scale_modifier=scale_modifier, transMat=EMPTY, view=cam.world_view_transform,
                proj=cam.full_proj_transform, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, H=H, W=W, sh=scene.shs, D=D,
                campos=cam.camera_center)


def cotangents(H, W, seed=1):
    rng = np.random.default_rng(seed)
    return rng.normal(size=(3, H, W)).astype(np.float32), rng.normal(size=(7, H, W)).astype(np.float32)


def run_oracle(oracle_mod, inp, grads=None):
    o = oracle_mod.Oracle()
    R, color, others, radii = o.rasterize_gaussians(
        inp["bg"], inp["means3D"], inp["colors"], inp["opacity"], inp["scales"], inp["rotations"],
        inp["scale_modifier"], inp["transMat"], inp["view"], inp["proj"], inp["tanfovx"], inp["tanfovy"], inp["H"],
        inp["W"], inp["sh"], inp["D"], inp["campos"])
    out = dict(R=R, color=color, others=others, radii=radii, oracle=o)
    if grads is not None:
        out["grads"] = o.rasterize_gaussians_backward(grads[0], grads[1])
    return out


def run_hip(inp, grads=None, debug=False, device="cuda:0"):
    """Calls the C ABI through the `_C` front-end exactly like the autograd node does."""
    import torch
    from g4splat_amd.diff_surfel_rasterization import _C
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=device)
    args = dict((k, t(v)) for k, v in inp.items() if isinstance(v, np.ndarray))
    fw = _C.rasterize_gaussians(args["bg"], args["means3D"], args["colors"], args["opacity"], args["scales"],
                                args["rotations"], inp["scale_modifier"], args["transMat"], args["view"], args["proj"],
                                inp["tanfovx"], inp["tanfovy"], inp["H"], inp["W"], args["sh"], inp["D"],
                                args["campos"], False, debug)
    R, color, others, radii, geom, binning, img = fw
    out = dict(R=R, color=color.cpu().numpy(), others=others.cpu().numpy(), radii=radii.cpu().numpy(), geom=geom,
               binning=binning, img=img, args=args)
    if grads is not None:
        g = _C.rasterize_gaussians_backward(args["bg"], args["means3D"], radii, args["colors"], args["scales"],
                                            args["rotations"], inp["scale_modifier"], args["transMat"], args["view"],
                                            args["proj"], inp["tanfovx"], inp["tanfovy"], t(grads[0]), t(grads[1]),
                                            args["sh"], inp["D"], args["campos"], geom, R, binning, img, debug)
        names = ("means2D", "colors", "opacity", "means3D", "transMat", "sh", "scales", "rotations")
        out["grads"] = dict((n, x.cpu().numpy()) for n, x in zip(names, g))
    return out


def hip_state(out, inp):
    """Reads the forward's private state back from the scratch chunks (g4s_rasterizer_layout)."""
    import ctypes
    from g4splat_amd import _lib
    L = _lib.G4sLayout()
    P = inp["means3D"].shape[0]
    rc = _lib.load().g4s_rasterizer_layout(P, int(out["R"]), inp["W"], inp["H"], ctypes.byref(L))
    assert rc == 0

    def chunk(t):
        a = t.cpu().numpy()
        off = (-t.data_ptr()) % 256  # the library aligns the chunk base to 256 B
        return a[off:]

    geom, binning, img = chunk(out["geom"]), chunk(out["binning"]), chunk(out["img"])
    N = inp["W"] * inp["H"]
    tiles = ((inp["W"] + 15) // 16) * ((inp["H"] + 15) // 16)
    R = int(out["R"])
    st = {}
    st["rec"] = geom[L.rec:L.rec + P * 128].view(np.float32).reshape(P, 32)
    st["rec_u32"] = geom[L.rec:L.rec + P * 128].view(np.uint32).reshape(P, 32)
    st["clamped"] = geom[L.clamped:L.clamped + P]
    st["depth_sorted"] = geom[L.depth_sorted:L.depth_sorted + 4 * P].view(np.uint32)
    st["tiles_touched"] = geom[L.tiles_touched:L.tiles_touched + 4 * P].view(np.uint32)
    st["ranges"] = img[L.ranges:L.ranges + 8 * tiles].view(np.uint32).reshape(tiles, 2)
    nb = int(st["tiles_touched"].sum())  # instances actually binned (<= R, the reference count)
    st["entries"] = binning[L.entries:L.entries + 8 * nb].view(np.uint64)
    st["qhit"] = binning[L.qhit:L.qhit + nb].copy()
    st["final_T"] = img[L.final_T:L.final_T + 12 * N].view(np.float32).reshape(3, N)
    st["n_contrib"] = img[L.n_contrib:L.n_contrib + 8 * N].view(np.uint32).reshape(2, N)
    st["tile_order"] = img[L.tile_order:L.tile_order + 4 * tiles].view(np.uint32)
    return st


def rel_err(a, b):
    """max |a-b| over max |b| (tensor-level relative error; SURVEY 8(d) 'grad rel-error')."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30)) if b.size else 0.0
