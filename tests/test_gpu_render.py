"""render() and the autograd operator on the MI355X vs the same host code driven by the CPU oracle."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from cpu_rasterizer import oracle_backend
from test_render_cpu import KEYS, _camera, _model
from g4splat_amd import synthetic
from g4splat_amd.gaussian_renderer import render

pytestmark = pytest.mark.gpu


def _to(cam, dev):
    d = dict(vars(cam))
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            d[k] = v.to(dev)
    return SimpleNamespace(**d)


def _loss(out):
    return (out["render"] ** 2).mean() + out["rend_dist"].mean() + (1 - (out["rend_normal"] * out["surf_normal"]).sum(0)).mean() \
        + 0.1 * out["surf_depth"].mean()


def test_render_matches_oracle_driven_render(hip_lib):
    cam, pipe = _camera(), SimpleNamespace(depth_ratio=0.5, compute_cov3D_python=False)
    ref_model, hip_model = _model(seed=2), _model(seed=2)
    with oracle_backend():
        ref = render(cam, ref_model, pipe, torch.tensor([0.1, 0.2, 0.3]))
    _loss(ref).backward()
    dev = torch.device("cuda:0")
    for name in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
        setattr(hip_model, name, torch.nn.Parameter(getattr(hip_model, name).detach().to(dev)))
    out = render(_to(cam, dev), hip_model, pipe, torch.tensor([0.1, 0.2, 0.3], device=dev))
    assert set(out) == KEYS
    _loss(out).backward()
    torch.cuda.synchronize()
    for k in KEYS - {"viewspace_points"}:
        a, b = out[k].detach().cpu(), ref[k].detach()
        if a.dtype.is_floating_point:
            assert (a - b).abs().max() <= 2e-4, k
        else:
            assert torch.equal(a, b), k
    for p, q in zip(hip_model.parameters(), ref_model.parameters()):
        g, r = p.grad.cpu(), q.grad
        assert (g - r).abs().max() <= 1e-3 * r.abs().max() + 1e-9
    g2, r2 = out["viewspace_points"].grad.cpu(), ref["viewspace_points"].grad
    assert (g2 - r2).abs().max() <= 1e-3 * r2.abs().max() + 1e-9


def test_no_grad_render_and_large_scene_properties(hip_lib):
    """Size-independent properties on a larger scene: alpha in [0,1], colour = blend + T*bg (linearity in
    bg), idempotence (same inputs -> bitwise same outputs)."""
    from common import run_hip, scene_inputs
    inp = scene_inputs(P=200000, W=640, H=480, seed=13, D=2, bg=(0.0, 0.0, 0.0), scale_mul=0.6)
    a = run_hip(inp)
    b = run_hip(inp)
    np.testing.assert_array_equal(a["color"], b["color"])
    np.testing.assert_array_equal(a["others"], b["others"])
    alpha = a["others"][1]
    assert alpha.min() >= 0 and alpha.max() <= 1 + 1e-6
    inp2 = dict(inp)
    inp2["bg"] = np.array([1.0, 0.5, 0.25], np.float32)
    c = run_hip(inp2)
    T = 1 - alpha
    for ch in range(3):
        assert np.abs(c["color"][ch] - (a["color"][ch] + T * inp2["bg"][ch])).max() <= 1e-5
    np.testing.assert_array_equal(c["others"], a["others"])
    assert (a["radii"] > 0).sum() > 1000


def test_metric_size_properties(hip_lib):
    """BASELINE configs[2] size (1.5 M surfels, 1600x1200, SH degree 3 -- bench.py's workload S3), through
    size-independent properties: idempotence (bitwise), sorted tile lists, every tile list in ascending depth,
    backward linear in the cotangents, invisible Gaussians get exactly zero gradient, num_rendered == sum of
    the reference's per-Gaussian tile-rect areas."""
    from common import EMPTY, hip_state
    from g4splat_amd import synthetic
    from g4splat_amd.diff_surfel_rasterization import _C
    P, W, H, D = 1_500_000, 1600, 1200, 3
    scene = synthetic.scene_room(P, seed=0)
    cam = synthetic.room_cameras(8, W, H, fovx_deg=90.0)[3]
    dev = "cuda:0"
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    empty = torch.empty(0, device=dev)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    args = dict(m=t(scene.means3D), o=t(scene.opacities), s=t(scene.scales), r=t(scene.rotations), sh=t(scene.shs),
                v=t(cam.world_view_transform), p=t(cam.full_proj_transform), c=t(cam.camera_center))

    def fwd():
        return _C.rasterize_gaussians(bg, args["m"], empty, args["o"], args["s"], args["r"], 1.0, empty, args["v"],
                                      args["p"], cam.tanfovx, cam.tanfovy, H, W, args["sh"], D, args["c"], False, False)

    def bwd(f, gc, go):
        R, color, others, radii, geom, binning, img = f
        return _C.rasterize_gaussians_backward(bg, args["m"], radii, empty, args["s"], args["r"], 1.0, empty, args["v"],
                                               args["p"], cam.tanfovx, cam.tanfovy, gc, go, args["sh"], D, args["c"],
                                               geom, R, binning, img, False)

    f1, f2 = fwd(), fwd()
    assert f1[0] == f2[0] and f1[0] > 4_000_000
    assert torch.equal(f1[1], f2[1]) and torch.equal(f1[2], f2[2]) and torch.equal(f1[3], f2[3])
    radii = f1[3]
    V = int((radii > 0).sum())
    assert 300_000 < V < 800_000
    alpha = f1[2][1]
    assert float(alpha.min()) >= 0 and float(alpha.max()) <= 1 + 1e-6 and torch.isfinite(f1[1]).all()

    # binning state: tile ids ascending over the whole list, depths ascending inside every tile range
    inp = dict(means3D=scene.means3D, W=W, H=H)
    st = hip_state(dict(R=f1[0], geom=f1[4], binning=f1[5], img=f1[6]), inp)
    ent = st["entries"]
    tile_of = (ent >> np.uint64(32)).astype(np.int64)
    assert np.all(np.diff(tile_of) >= 0)
    view = np.asarray(cam.world_view_transform, np.float32)
    z = scene.means3D @ view[:3, 2] + view[3, 2]  # p_view.z
    zi = z[(ent & np.uint64(0xFFFFFFFF)).astype(np.int64)]
    same_tile = np.diff(tile_of) == 0
    assert np.all(np.diff(zi)[same_tile] >= -1e-6)
    rng = st["ranges"]
    assert int((rng[:, 1] - rng[:, 0]).sum()) == len(ent)

    # backward: linear in the cotangents, zero rows for invisible Gaussians, bitwise reproducible
    g = torch.Generator(device=dev).manual_seed(5)
    gc1, go1 = torch.randn((3, H, W), device=dev, generator=g), torch.randn((7, H, W), device=dev, generator=g)
    gc2, go2 = torch.randn((3, H, W), device=dev, generator=g), torch.randn((7, H, W), device=dev, generator=g)
    ga, gb = bwd(f1, gc1, go1), bwd(f1, gc2, go2)
    gab = bwd(f1, 2.0 * gc1 - 0.5 * gc2, 2.0 * go1 - 0.5 * go2)
    ga2 = bwd(f1, gc1, go1)
    names = ("means2D", "colors", "opacity", "means3D", "transMat", "sh", "scales", "rotations")
    invisible = radii <= 0
    for n, x, y, xy, x2 in zip(names, ga, gb, gab, ga2):
        assert torch.equal(x, x2), n
        assert torch.isfinite(x).all(), n
        if x.numel() == 0:
            continue
        lin = 2.0 * x - 0.5 * y
        scale = float(lin.abs().max())
        assert float((xy - lin).abs().max()) <= 2e-4 * scale + 1e-9, n
        assert not x[invisible].any(), n


def test_pack_rows_roundtrip(hip_lib):
    """g4s_pack_rows (visible-rows gradient exchange): pack == torch index_select per segment, unpack restores."""
    import ctypes
    P, widths = 5000, [3, 48, 1, 2, 4, 2]
    g = torch.Generator(device="cuda:0").manual_seed(3)
    segs = [torch.randn((P, w), device="cuda:0", generator=g) for w in widths]
    idx = (torch.rand(P, device="cuda:0", generator=g) < 0.3).nonzero(as_tuple=True)[0]
    n = int(idx.numel())
    packed = torch.full((n * sum(widths),), float("nan"), device="cuda:0")
    ptrs = (ctypes.c_void_p * len(segs))(*[s.data_ptr() for s in segs])
    wid = (ctypes.c_int * len(segs))(*widths)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert hip_lib.g4s_pack_rows(len(segs), ptrs, wid, ctypes.c_void_p(idx.data_ptr()), n,
                                 ctypes.c_void_p(packed.data_ptr()), 0, stream) == 0
    want = torch.cat([s.index_select(0, idx).reshape(-1) for s in segs])
    assert torch.equal(packed, want)
    before = [s.clone() for s in segs]
    packed *= 2.0
    assert hip_lib.g4s_pack_rows(len(segs), ptrs, wid, ctypes.c_void_p(idx.data_ptr()), n,
                                 ctypes.c_void_p(packed.data_ptr()), 1, stream) == 0
    torch.cuda.synchronize()
    keep = torch.ones(P, dtype=torch.bool, device="cuda:0")
    keep[idx] = False
    for s, b in zip(segs, before):
        assert torch.equal(s[idx], 2.0 * b[idx]) and torch.equal(s[keep], b[keep])
    assert hip_lib.g4s_pack_rows(9, ptrs, wid, None, 0, None, 0, stream) < 0
    # row-major buffer [n, sum w] (mode bit 1) and accumulating unpack (bit 2): what OwnerReduce moves through all_to_all
    rm = torch.full((n, sum(widths)), float("nan"), device="cuda:0")
    assert hip_lib.g4s_pack_rows(len(segs), ptrs, wid, ctypes.c_void_p(idx.data_ptr()), n,
                                 ctypes.c_void_p(rm.data_ptr()), 2, stream) == 0
    assert torch.equal(rm, torch.cat([s.index_select(0, idx) for s in segs], dim=1))
    before = [s.clone() for s in segs]
    assert hip_lib.g4s_pack_rows(len(segs), ptrs, wid, ctypes.c_void_p(idx.data_ptr()), n,
                                 ctypes.c_void_p(rm.data_ptr()), 7, stream) == 0
    torch.cuda.synchronize()
    for s, b in zip(segs, before):
        assert torch.equal(s[idx], b[idx] + b[idx]) and torch.equal(s[keep], b[keep])
    rm.fill_(1.5)
    assert hip_lib.g4s_pack_rows(len(segs), ptrs, wid, ctypes.c_void_p(idx.data_ptr()), n,
                                 ctypes.c_void_p(rm.data_ptr()), 3, stream) == 0
    torch.cuda.synchronize()
    for s, b in zip(segs, before):
        assert bool((s[idx] == 1.5).all()) and torch.equal(s[keep], b[keep])
    assert hip_lib.g4s_pack_rows(len(segs), ptrs, wid, ctypes.c_void_p(idx.data_ptr()), n,
                                 ctypes.c_void_p(rm.data_ptr()), 4, stream) < 0  # add without unpack
    # mode bit 3: a trailing int32 index column -- rows and their indices travel in ONE all_to_all (OwnerReduce): pack
    # writes it, the accumulating unpack reads the row indices from the buffer (row_index = NULL)
    W = sum(widths)
    rc = torch.full((n, W + 1), float("nan"), device="cuda:0")
    assert hip_lib.g4s_pack_rows(len(segs), ptrs, wid, ctypes.c_void_p(idx.data_ptr()), n,
                                 ctypes.c_void_p(rc.data_ptr()), 10, stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(rc[:, :W], torch.cat([s.index_select(0, idx) for s in segs], dim=1))
    assert torch.equal(rc[:, W].contiguous().view(torch.int32).to(torch.int64), idx)
    before = [s.clone() for s in segs]
    assert hip_lib.g4s_pack_rows(len(segs), ptrs, wid, None, n, ctypes.c_void_p(rc.data_ptr()), 15, stream) == 0
    torch.cuda.synchronize()
    for s, b in zip(segs, before):
        assert torch.equal(s[idx], b[idx] + b[idx]) and torch.equal(s[keep], b[keep])
    assert hip_lib.g4s_pack_rows(len(segs), ptrs, wid, None, n, ctypes.c_void_p(rc.data_ptr()), 9, stream) < 0  # needs row-major
    assert hip_lib.g4s_pack_rows(len(segs), ptrs, wid, None, n, ctypes.c_void_p(rc.data_ptr()), 10, stream) < 0  # pack needs idx


@pytest.mark.parametrize("D,M", [(3, 16), (1, 16), (0, 16), (2, 9), (0, 1)])
def test_split_sh_is_bitwise_the_packed_path(hip_lib, D, M):
    """g4s_rasterizer_*_split_sh (features_dc / features_rest read in place, gradients written separately) against
    the packed [P,M,3] entry points: identical images, radii and gradients, bit for bit."""
    from common import scene_inputs
    from g4splat_amd.diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    inp = scene_inputs(P=30000, W=320, H=208, seed=40 + D, D=3, bg=(0.2, 0.1, 0.3), scale_mul=0.8)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    sh_all = t(inp["sh"])[:, :M, :].contiguous()
    rs = GaussianRasterizationSettings(
        image_height=inp["H"], image_width=inp["W"], tanfovx=inp["tanfovx"], tanfovy=inp["tanfovy"], bg=t(inp["bg"]),
        scale_modifier=1.0, viewmatrix=t(inp["view"]), projmatrix=t(inp["proj"]), sh_degree=D,
        campos=t(inp["campos"]), prefiltered=False, debug=False)
    gen = torch.Generator(device="cpu").manual_seed(5)
    w_c = torch.randn((3, inp["H"], inp["W"]), generator=gen).to(dev)
    w_o = torch.randn((7, inp["H"], inp["W"]), generator=gen).to(dev)

    def run(split):
        leaves = {k: t(inp[k]).requires_grad_(True) for k in ("means3D", "opacity", "scales", "rotations")}
        m2 = torch.zeros_like(leaves["means3D"], requires_grad=True)
        if split:
            dc = sh_all[:, :1, :].clone().requires_grad_(True)
            rest = sh_all[:, 1:, :].clone().requires_grad_(True)
            shs = (dc, rest)
        else:
            packed = sh_all.clone().requires_grad_(True)
            shs = packed
        color, radii, others = GaussianRasterizer(rs)(means3D=leaves["means3D"], means2D=m2, opacities=leaves["opacity"],
                                                      shs=shs, scales=leaves["scales"], rotations=leaves["rotations"])
        ((color * w_c).sum() + (others * w_o).sum()).backward()
        g_sh = torch.cat((dc.grad, rest.grad), dim=1) if split else packed.grad
        return color.detach(), radii, others.detach(), g_sh, m2.grad, {k: v.grad for k, v in leaves.items()}

    a, b = run(False), run(True)
    torch.cuda.synchronize()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    assert a[3].abs().max() > 0
    if (D + 1) ** 2 < M:
        assert a[3][:, (D + 1) ** 2:, :].abs().max() == 0  # coefficients above the active degree get zero gradient
    for k in a[5]:
        assert torch.equal(a[5][k], b[5][k]), k


def test_split_sh_argument_errors(hip_lib):
    from g4splat_amd.diff_surfel_rasterization import _C
    dev = torch.device("cuda:0")
    z = lambda *s: torch.zeros(s, device=dev)
    bad = (z(10, 3), z(10, 15, 3))  # features_dc must be [P,1,3]
    with pytest.raises(RuntimeError, match="split SH"):
        _C.rasterize_gaussians(z(3), z(10, 3), z(0), z(10, 1), z(10, 2), z(10, 4), 1.0, z(0), z(4, 4), z(4, 4), 1.0, 1.0,
                               32, 32, bad, 3, z(3), False, False)


def test_more_than_four_million_emitting_gaussians(hip_lib):
    """Above 4.3 M keys the depth sort's chunks stop growing (2 048 keys) and their number does instead: the
    histogram capacity has to follow.  5.5 M small visible surfels, rendered twice with the Gaussians in a different
    memory order: the depth sort makes the image independent of that order, bit for bit (all depths distinct)."""
    from g4splat_amd.diff_surfel_rasterization import _C
    dev = torch.device("cuda:0")
    P, W, H = 5_500_000, 512, 512
    g = torch.Generator(device="cpu").manual_seed(77)
    # distinct depths (random floats collide by the hundred thousand at this count, and ties resolve by index)
    z = (1.0 + 8.0 * (torch.randperm(P, generator=g).double() + 0.5) / P).float()
    assert torch.unique(z).numel() == P
    xy = (torch.rand((P, 2), generator=g) - 0.5) * 1.1 * z[:, None]
    means = torch.cat([xy, z[:, None]], 1)
    scales = torch.full((P, 2), 0.004)
    rots = torch.nn.functional.normalize(torch.randn((P, 4), generator=g))
    opac = torch.rand((P, 1), generator=g) * 0.2 + 0.02
    cols = torch.rand((P, 3), generator=g)
    eye = torch.eye(4)
    tan = 0.5774
    proj = torch.tensor([[1 / tan, 0, 0, 0], [0, 1 / tan, 0, 0], [0, 0, 100.0 / 99.99, 1], [0, 0, -0.01 * 100.0 / 99.99, 0]])
    e = torch.empty(0, device=dev)

    def run(order):
        t = lambda a: a[order].contiguous().to(dev)
        out = _C.rasterize_gaussians(torch.zeros(3, device=dev), t(means), t(cols), t(opac), t(scales), t(rots), 1.0, e,
                                     eye.to(dev), proj.to(dev), tan, tan, H, W, e, 0, torch.zeros(3, device=dev), False, False)
        return out[0], out[1].cpu(), out[2].cpu(), out[3].cpu()

    ident = torch.arange(P)
    perm = torch.randperm(P, generator=g)
    Ra, ca, oa, ra = run(ident)
    Rb, cb, ob, rb = run(perm)
    assert Ra == Rb and int((ra > 0).sum()) > 4_400_000
    assert torch.equal(ra[perm], rb)
    assert torch.equal(ca, cb) and torch.equal(oa, ob)
    assert float(oa[1].max()) > 0.5  # the frame is really covered


def test_wrong_element_counts_are_refused(hip_lib):
    """Raw pointers go to the native side: tensors with the wrong number of elements (3-D Gaussian splatting's [P,3]
    scales, a [3,3] view matrix, ...) are refused with a message instead of being read out of bounds."""
    from g4splat_amd.diff_surfel_rasterization import _C
    dev = torch.device("cuda:0")
    P = 50
    z = lambda *s: torch.zeros(s, device=dev)
    e = torch.empty(0, device=dev)
    ok = dict(bg=z(3), m=z(P, 3), col=e, opa=z(P, 1), sc=z(P, 2), rot=z(P, 4), tm=e, view=torch.eye(4, device=dev),
              proj=torch.eye(4, device=dev), sh=z(P, 16, 3), cam=z(3))

    def call(**over):
        a = dict(ok, **over)
        return _C.rasterize_gaussians(a["bg"], a["m"], a["col"], a["opa"], a["sc"], a["rot"], 1.0, a["tm"], a["view"], a["proj"],
                                      1.0, 1.0, 32, 32, a["sh"], 3, a["cam"], False, False)

    call()  # the well-formed call goes through
    for over, msg in ((dict(sc=z(P, 3)), "scales"), (dict(rot=z(P, 3)), "rotations"), (dict(opa=z(P - 1, 1)), "opacity"),
                      (dict(view=torch.eye(3, device=dev)), "viewmatrix"), (dict(bg=z(4)), "background"),
                      (dict(sh=z(P, 16)), "sh must have"), (dict(sh=z(P + 1, 16, 3)), "sh must have"),
                      (dict(cam=z(2)), "campos")):
        with pytest.raises(RuntimeError, match=msg):
            call(**over)


def test_scene_file_in_the_reference_layout_loads_and_renders(hip_lib):
    """tests/golden/scene_ref_layout.ply (the reference's on-disk format, packed byte by byte by
    tests/golden/make_golden_ply.py -- no code shared with ply_io.py; 200 surfels, SH degree 3, mip filter) is loaded
    with GaussianModel.load_ply straight onto the GPU and rendered through render(); the result must equal the oracle-
    driven render of the same file loaded on the host."""
    import os
    from g4splat_amd.gaussian_model import GaussianModel
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scene_ref_layout.ply")
    dev = torch.device("cuda:0")
    host, gpu = GaussianModel(3), GaussianModel(3)
    host.load_ply(path, device="cpu")
    gpu.load_ply(path)  # default device: "cuda", like the reference (:484-491)
    assert gpu._xyz.is_cuda and gpu._xyz.shape == (200, 3) and gpu.use_mip_filter
    W, H = 160, 112
    c = synthetic.look_at_camera((-0.8, 0.0, -0.4), (1.0, 0.1, 0.6), (0, 1, 0), 1.9, W, H)  # inside the 2 x 1.5 x 1.2 m box
    cam = SimpleNamespace(image_width=W, image_height=H, FoVx=c.FoVx, FoVy=c.FoVy, znear=0.01, zfar=100.0,
                          world_view_transform=torch.tensor(c.world_view_transform),
                          full_proj_transform=torch.tensor(c.full_proj_transform), camera_center=torch.tensor(c.camera_center))
    pipe = SimpleNamespace(depth_ratio=1.0, compute_cov3D_python=False)
    bg = torch.tensor([0.2, 0.3, 0.1])
    with oracle_backend():
        ref = render(cam, host, pipe, bg)
    out = render(_to(cam, dev), gpu, pipe, bg.to(dev))
    assert int(ref["visibility_filter"].sum()) > 50
    for k in KEYS - {"viewspace_points"}:
        a, b = out[k].detach().cpu(), ref[k].detach()
        if a.dtype.is_floating_point:
            # (5e-5: rend_depth = depth sum / alpha sum amplifies the rasterizer's ~3e-7 absolute differences a hundredfold
            # where a pixel is 1 % covered; measured 2.3e-5 there since the affine ray-splat form of round 5, 1e-5 before)
            assert (a - b).abs().max() <= 5e-5, (k, float((a - b).abs().max()))
        else:
            assert torch.equal(a, b), k
