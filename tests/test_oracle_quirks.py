"""Explicit assertions of the reference's non-gradient quirks and tie/clipping rules, on the oracle
(SURVEY.md 8(a) a12-a15, 8(c)).  The HIP path is then held to the oracle by the -m gpu tests."""
import math

import numpy as np

from common import EMPTY, cotangents, run_oracle, scene_inputs


def _tiny(P=40, W=64, H=48, seed=4, **kw):
    inp = scene_inputs(P=P, W=W, H=H, seed=seed, D=1, scale_mul=4.0, **kw)
    return inp


def test_backward_ignores_scale_modifier(oracle_mod):
    """backward.cu:481 builds S with scale_to_mat(scale, 1.0f): with scale_modifier != 1 the T-path
    gradients are those of the *unmodified* scales evaluated with the forward's blend terms."""
    a = _tiny(scale_modifier=1.0)
    b = dict(a)
    b["scale_modifier"] = 0.5
    b["scales"] = (a["scales"] * 2.0).astype(np.float32)  # same effective extent in the forward
    g = cotangents(48, 64)
    oa, ob = run_oracle(oracle_mod, a, g), run_oracle(oracle_mod, b, g)
    np.testing.assert_allclose(oa["color"], ob["color"], atol=1e-6)
    # identical forward, but backward sees scales twice as large in b: the raw T gradient is the same ...
    np.testing.assert_allclose(oa["oracle"].state("dL_dtransMat_raw"), ob["oracle"].state("dL_dtransMat_raw"),
                               rtol=1e-4, atol=1e-4)
    # ... and dL_dscale is NOT rescaled by the modifier (a true gradient would be 0.5x)
    # (Gaussians whose low-pass branch fired are left out: their centre fold uses the backward's own T)
    vis = (oa["radii"] > 0) & (np.abs(oa["oracle"].state("dL_dmean2D_raw")).sum(1) == 0)
    assert vis.sum() >= 5
    ga, gb = oa["grads"]["scales"][vis], ob["grads"]["scales"][vis]
    big = np.abs(ga) > 1e-3 * np.abs(ga).max()
    np.testing.assert_allclose(gb[big], ga[big], rtol=2e-3)


def test_backward_width_height_truncation(oracle_mod):
    """backward.cu:618-619: W = int(focal_x * tan_fovx * 2) can come out as W - 1 in float32."""
    hits = 0
    for W in range(50, 400, 7):
        for fov in np.linspace(0.4, 1.9, 23):
            t = np.float32(math.tan(fov / 2))
            focal = np.float32(W) / (np.float32(2.0) * t)
            if int(np.float32(focal * t) * np.float32(2)) != W:
                hits += 1
    assert hits > 0  # the quirk is real in float32; the oracle / HIP code restates the expression verbatim


def test_alpha_clamp_gradient_is_not_gated(oracle_mod):
    """backward.cu:390: dL_dG = opacity * dL_dalpha even when alpha was clamped to 0.99."""
    inp = scene_inputs(P=1, W=32, H=32, seed=0, D=0)
    inp["means3D"] = np.array([[0.0, 0.0, 2.0]], np.float32)
    inp["scales"] = np.array([[0.5, 0.5]], np.float32)
    inp["rotations"] = np.array([[1, 0, 0, 0]], np.float32)
    inp["opacity"] = np.array([[1.0]], np.float32)  # alpha = min(0.99, 1.0 * G) is clamped near the centre
    gc = np.ones((3, 32, 32), np.float32)
    o = run_oracle(oracle_mod, inp, (gc, np.zeros((7, 32, 32), np.float32)))
    assert np.abs(o["grads"]["opacity"]).max() > 0
    assert np.abs(o["grads"]["scales"]).max() > 0  # a gated gradient would vanish where alpha == 0.99


def test_mean2d_output_is_densification_surrogate(oracle_mod):
    """backward.cu:637-640: dL_dmean2D.xy = dL_dT[2|5] * T[8] * 0.5 * (W|H), z = 0."""
    inp = _tiny()
    g = cotangents(48, 64)
    o = run_oracle(oracle_mod, inp, g)
    orc = o["oracle"]
    vis = o["radii"] > 0
    raw, T = orc.state("dL_dtransMat_raw"), orc.state("transMat")
    f = lambda v: np.float32(v)
    Wb = int(f(f(64) / (f(2.0) * f(inp["tanfovx"]))) * f(inp["tanfovx"]) * f(2))
    Hb = int(f(f(48) / (f(2.0) * f(inp["tanfovy"]))) * f(inp["tanfovy"]) * f(2))
    want_x = (raw[:, 2] * T[:, 8]).astype(np.float64) * 0.5 * Wb
    want_y = (raw[:, 5] * T[:, 8]).astype(np.float64) * 0.5 * Hb
    np.testing.assert_allclose(o["grads"]["means2D"][vis, 0], want_x[vis].astype(np.float32), rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(o["grads"]["means2D"][vis, 1], want_y[vis].astype(np.float32), rtol=1e-6, atol=1e-12)
    assert not np.any(o["grads"]["means2D"][:, 2])
    assert not np.any(o["grads"]["means2D"][~vis])


def test_equal_depth_ties_resolve_by_index(oracle_mod):
    """Stable sort (rasterizer_impl.cu:304-309): equal depth bits inside a tile keep ascending index."""
    inp = scene_inputs(P=5, W=32, H=32, seed=0, D=0)
    inp["means3D"] = np.array([[0.0, 0, 2.0]] * 5, np.float32) + np.array([[0.01 * i, 0, 0] for i in range(5)], np.float32)
    inp["scales"] = np.full((5, 2), 0.2, np.float32)
    inp["rotations"] = np.tile(np.array([[1, 0, 0, 0]], np.float32), (5, 1))
    inp["opacity"] = np.full((5, 1), 0.5, np.float32)
    o = run_oracle(oracle_mod, inp)
    pl, rg = o["oracle"].state("point_list"), o["oracle"].state("ranges")
    for a, b in rg:
        assert np.all(np.diff(pl[a:b].astype(np.int64)) > 0)


def test_tile_rect_clips_opaque_splats(oracle_mod):
    """auxiliary.h:66-76: the 3-sigma rect decides which 16-px tiles see a splat, so an opaque splat
    (alpha at 3 sigma = 0.011 * opacity > 1/255) is visibly cut where the rect ends on a tile border."""
    worst = 0.0
    for dx in np.linspace(-0.2, 0.2, 17):
        inp = scene_inputs(P=1, W=96, H=96, seed=0, D=0)
        inp["means3D"] = np.array([[dx, 0.0, 2.0]], np.float32)
        inp["scales"] = np.array([[0.17, 0.17]], np.float32)
        inp["rotations"] = np.array([[1, 0, 0, 0]], np.float32)
        inp["opacity"] = np.array([[0.99]], np.float32)
        o = run_oracle(oracle_mod, inp)
        alpha = o["others"][1]
        rg = o["oracle"].state("ranges").reshape(6, 6, 2)
        covered = (rg[:, :, 1] - rg[:, :, 0]) > 0
        assert not np.any(alpha[~np.kron(covered, np.ones((16, 16), bool))])  # nothing outside the rect
        for ty in range(6):
            for tx in range(6):
                if not covered[ty, tx]:
                    continue
                blk = alpha[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16]
                for (dy, dxx, sl) in ((0, 1, blk[:, -1]), (0, -1, blk[:, 0]), (1, 0, blk[-1, :]), (-1, 0, blk[0, :])):
                    ny, nx = ty + dy, tx + dxx
                    if 0 <= ny < 6 and 0 <= nx < 6 and not covered[ny, nx]:
                        worst = max(worst, float(sl.max()))
    # for some placement the last covered pixel column still carries clearly visible alpha
    assert worst > 2.0 / 255.0


def test_median_contributor_bookkeeping(oracle_mod):
    """forward.cu:318,404-408,429: no pixel with T always <= 0.5 ... median stays 0 -> n_contrib[1] = 0."""
    inp = _tiny()
    o = run_oracle(oracle_mod, inp)
    nc = o["oracle"].state("n_contrib")
    med_depth = o["others"][5].reshape(-1)
    assert np.all((nc[1] == 0) == (med_depth == 0))
    assert np.all(nc[1] <= nc[0])


def test_higher_msb(oracle_mod):
    for n, want in ((1, 1), (2, 2), (117, 7), (256, 9), (7500, 13), (65535, 16)):
        assert oracle_mod.get_higher_msb(n) == want


def test_backward_is_a_sum_of_per_pixel_contributions(oracle_mod):
    """What the parity gate's masked re-run relies on (tests/common.py): the backward is linear in the cotangents and a
    pixel's contribution depends on that pixel's cotangents only -- zeroing the cotangents of a pixel set removes
    exactly that set's contribution, and the oracle object can run several backwards over one forward."""
    from common import cotangents, run_oracle, scene_inputs
    inp = scene_inputs(P=400, W=48, H=40, seed=5, D=2)
    g = cotangents(40, 48, seed=2)
    o = run_oracle(oracle_mod, inp, g)
    rng = np.random.default_rng(0)
    mask = rng.random((40, 48)) < 0.3
    ga = (np.where(mask, 0, g[0]).astype(np.float32), np.where(mask, 0, g[1]).astype(np.float32))
    gb = (np.where(mask, g[0], 0).astype(np.float32), np.where(mask, g[1], 0).astype(np.float32))
    a = o["oracle"].rasterize_gaussians_backward(*ga)
    b = o["oracle"].rasterize_gaussians_backward(*gb)
    again = o["oracle"].rasterize_gaussians_backward(*g)
    for k in o["grads"]:
        np.testing.assert_array_equal(again[k], o["grads"][k])
        full = o["grads"][k].astype(np.float64)
        np.testing.assert_allclose(a[k].astype(np.float64) + b[k].astype(np.float64), full,
                                   atol=2e-5 * (np.abs(full).max() + 1e-30), rtol=0)


def test_pixel_alternatives_restate_the_forward_loop():
    """oracle.pixel_alternatives (the parity gate's check on threshold-flipped pixels, tests/common.py): slot 0 -- no
    decision flipped -- is the frame as rendered, bit for bit, contributor ids included; with a margin so wide that
    every decision counts as "close", other slots differ from it (a blended splat skipped, an early stop, ...)."""
    import numpy as np
    from common import run_oracle, scene_inputs, _last_and_median_ids
    from oracle import oracle as om

    inp = scene_inputs(P=3000, W=96, H=64, seed=21, D=2, bg=(0.2, 0.5, 0.1), scale_mul=2.0)
    o = run_oracle(om, inp)
    orc = o["oracle"]
    N = 96 * 64
    pix = np.arange(0, N, 7)
    alt = om.pixel_alternatives(orc, pix, 1e-5)
    want = np.concatenate([o["color"].reshape(3, N), o["others"].reshape(7, N)], 0)[:, pix].T
    np.testing.assert_array_equal(alt[:, 0, :10].astype(np.float32), want)
    ids = _last_and_median_ids(orc.state("n_contrib"), orc.state("ranges"), orc.state("point_list"), 96, 64)
    np.testing.assert_array_equal(alt[:, 0, 10], ids[0][pix])
    np.testing.assert_array_equal(alt[:, 0, 11], ids[1][pix])
    # nothing is within 1e-5 of a threshold in most pixels: all slots equal slot 0 there
    margins = om.pixel_margins(orc).min(axis=0)[pix]
    far = margins > 1e-5
    assert far.sum() > 0.9 * len(pix)
    assert np.all(alt[far] == alt[far][:, :1])
    wide = om.pixel_alternatives(orc, pix, 1e30)
    covered = want[:, 4] > 0.05  # pixels something was blended into
    # (only the first four decisions of a walk can be flipped, and most of those concern splats that miss the pixel anyway)
    assert np.any(wide[covered][:, 1:, :10] != wide[covered][:, :1, :10], axis=(1, 2)).mean() > 0.3


def test_skip_suspects_are_the_pixels_of_the_skip_margin_map():
    """oracle.skip_suspects lists every (pixel, Gaussian) pair whose skip decision is within the margin of its threshold:
    its pixels are exactly those where oracle.pixel_margins' skip row is below the margin, and every listed Gaussian is on
    that pixel's tile list."""
    import numpy as np
    from common import run_oracle, scene_inputs
    from oracle import oracle as om

    inp = scene_inputs(P=3000, W=96, H=64, seed=21, D=2, bg=(0.2, 0.5, 0.1), scale_mul=2.0)
    orc = run_oracle(om, inp)["oracle"]
    for margin in (1e-3, 5e-2):
        pix, gid = om.skip_suspects(orc, margin)
        want = np.nonzero(om.pixel_margins(orc)[0] < margin)[0]
        assert len(pix) >= len(want) > 0
        np.testing.assert_array_equal(np.unique(pix), want)
        plist, rng = orc.state("point_list"), orc.state("ranges")
        for p, g in list(zip(pix, gid))[:200]:
            t = (p // 96 // 16) * 6 + (p % 96) // 16
            assert g in plist[rng[t, 0]:rng[t, 1]]


def test_skip_suspects_margins_equal_pixel_margins():
    """The parity gate takes its threshold margins from skip_suspects' walk (one pass over the frame for both): they must
    be oracle.pixel_margins', bit for bit."""
    import numpy as np
    from common import run_oracle, scene_inputs
    from oracle import oracle as om

    inp = scene_inputs(P=3000, W=96, H=64, seed=22, D=1, scale_mul=2.0)
    orc = run_oracle(om, inp)["oracle"]
    _pix, _gid, m = om.skip_suspects(orc, 1e-3, with_margins=True)
    np.testing.assert_array_equal(m, om.pixel_margins(orc))
