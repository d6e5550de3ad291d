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
        out["cot"] = grads
    return out


def run_hip(inp, grads=None, debug=False, device="cuda:0", backend=None):
    """Calls the C ABI through the `_C` front-end exactly like the autograd node does.  `backend`: another module with
    the same surface (the compiled pybind stub of INTEGRATION.md section B) instead of the ctypes front-end."""
    import torch
    from g4splat_amd.diff_surfel_rasterization import _C
    if backend is not None:
        _C = backend
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
        out["cot"] = grads
        out["debug"] = debug
    return out


def hip_backward_again(h, inp, grads, device="cuda:0"):
    """A second backward over the forward state of `h` (run_hip) with other cotangents."""
    import torch
    from g4splat_amd.diff_surfel_rasterization import _C
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=device)
    args = h["args"]
    g = _C.rasterize_gaussians_backward(args["bg"], args["means3D"], t(h["radii"]), args["colors"], args["scales"],
                                        args["rotations"], inp["scale_modifier"], args["transMat"], args["view"],
                                        args["proj"], inp["tanfovx"], inp["tanfovy"], t(grads[0]), t(grads[1]),
                                        args["sh"], inp["D"], args["campos"], h["geom"], h["R"], h["binning"], h["img"],
                                        h.get("debug", False))
    names = ("means2D", "colors", "opacity", "means3D", "transMat", "sh", "scales", "rotations")
    return dict((n, x.cpu().numpy()) for n, x in zip(names, g))


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
    st["hot_count"] = int(img[L.hot_count:L.hot_count + 4].view(np.uint32)[0])  # (of the last backward, if one ran)
    return st


def rel_err(a, b):
    """max |a-b| over max |b| (tensor-level relative error; SURVEY 8(d) 'grad rel-error')."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30)) if b.size else 0.0


# ---------------------------------------------------------------------------------------------------------
# Parity bars and the threshold-margin proof.
#
# CONTRACT bars (BASELINE.json north_star): outputs <= 1e-4 abs, gradients <= 1e-3 rel (tensor-level).
# GUARD bars: ~10x what the HIP path actually achieves against the oracle (measured on MI355X with
# tools/parity_report.py, profiles/r02_parity_report.txt): they are what the tests assert, so that a regression
# of one order of magnitude is caught long before the contract is at risk.
#
# (max_flipped_frac: until round 4 the HIP path shared the oracle's arithmetic up to one v_rcp_f32 and one v_exp_f32 per pair
# (~3e-7 relative in alpha) and flipped 9e-6 of S3's pixels -- that gate, cap 2e-5, still holds for option `no_fastpath`
# (STRICT below).  The affine ray-splat intersection of round 5 (csrc/g4s_device.h) is a different, equally accurate
# single-precision evaluation of the same quantity, ~1e-6 relative from the oracle's, and flips twice as many of the
# pixels that sit on a threshold: 1.2e-5 .. 1.8e-5 of S3's frames (profiles/r06_affine_tol.txt), cap 4e-5.)
# A pixel or gradient row outside a bar is accepted ONLY if it is *explained*: the oracle reports, per pixel,
# how close each discrete decision of the forward loop (alpha >= 1/255, depth >= near, T(1-alpha) >= 1e-4,
# T > 0.5) came to its threshold (oracle.pixel_margins).  Any two correct single-precision evaluations differ
# by ~1e-6 relative there, so within MARGIN either side is a correct answer.  Everything else fails the test.
OUT_ATOL = 1e-4          # contract
GRAD_RTOL = 1e-3         # contract, max|a-b| / max|b| per tensor
OUT_ATOL_GUARD = 2e-5    # guard: measured <= 2.9e-6 on S1..S5 and 60 fuzz scenes (worst map: the depth sum, values up to ~6)
GRAD_RTOL_GUARD = 1e-4   # guard: measured <= 1.1e-5 tensor-level (typically 1e-6)
ROW_RTOL_GUARD = 1e-2    # guard, row-level (measured <= 8.6e-4): |a-b|_row,inf / |b|_row,inf for rows above ROW_FLOOR of the tensor's max
ROW_FLOOR = 1e-3
MARGIN = 1e-5            # relative distance to a decision threshold below which either side is correct.  Round 4's value
                         # again (round 5 had tied it to AFFINE_TOL = 4e-5): the sweep of profiles/r06_affine_tol.txt shows every
                         # flipped pixel of S1 / S3 / S3t within 1e-5 of its threshold whatever AFFINE_TOL (1e-5, 2e-5, 4e-5) --
                         # the gate is independent of the certificate's own bound (now 2e-5) again
CONTRACT_PIXELS_FRAC = 4e-5   # pixels that may lie beyond the contract's 1e-4 abs against the oracle (threshold flips), of the frame
STRICT = dict(max_flipped_frac=2e-5, exempt_skip_suspects=False)   # round 4's gate, for runs on the reference's arithmetic


def _last_and_median_ids(n_contrib, ranges, ids, W, H):
    """Per pixel: Gaussian id of the last contributor and of the median contributor (-1 = none).  List positions
    differ between the two sides (the HIP lists are culled subsequences of the reference lists), ids do not."""
    ty, tx = np.mgrid[0:H, 0:W]
    tile = ((ty // 16) * ((W + 15) // 16) + tx // 16).reshape(-1)
    out = []
    for row in (0, 1):
        c = n_contrib.reshape(2, -1)[row].astype(np.int64)
        pos = ranges[tile, 0].astype(np.int64) + c - 1
        out.append(np.where(c > 0, ids[np.clip(pos, 0, max(len(ids) - 1, 0))].astype(np.int64) if len(ids) else -1, -1))
    return out


def parity_report(h, o, inp, oracle_mod, scale_aware=False, masked_rerun=True, exempt_skip_suspects=True):
    """Compares a HIP result `h` (run_hip) with the oracle's `o` (run_oracle) and classifies every mismatch.
    Returns a dict of measured errors (over the unexplained part) and of the explained sets.
    scale_aware: output errors of a map are taken relative to max(1, max |map|) -- for scenes blown up on purpose
    (tools/fuzz_sweep.py FUZZ_SCENE_SCALE), whose depth maps hold values of 10^2..10^3 with a float ulp above the bar."""
    W, H = inp["W"], inp["H"]
    N = W * H
    orc = o["oracle"]
    rep = dict(N=N, R=int(o["R"]))
    sk_pix, sk_gid, margins = oracle_mod.skip_suspects(orc, MARGIN, with_margins=True)  # margins [3, N]: one walk for both
    suspect = margins.min(axis=0) < MARGIN           # either side of some threshold is a correct answer here
    rep["suspect_pixels"] = int(suspect.sum())
    # ---- outputs
    diffs = np.concatenate([np.abs(h["color"] - o["color"]), np.abs(h["others"] - o["others"])], 0).reshape(10, N)
    if scale_aware and N:
        mag = np.concatenate([np.abs(o["color"]), np.abs(o["others"])], 0).reshape(10, N).max(axis=1)
        diffs = diffs / np.maximum(1.0, mag)[:, None]
    dmax = diffs.max(axis=0)
    rep["out_err_unexplained"] = float(dmax[~suspect].max()) if (~suspect).any() else 0.0
    rep["out_err_per_map_unexplained"] = [float(d[~suspect].max()) if (~suspect).any() else 0.0 for d in diffs]
    rep["out_err_all"] = float(dmax.max()) if N else 0.0
    # the contract read literally (north_star: "outputs <= 1e-4 abs"): how many pixels lie beyond it, and the worst of them
    rep["pixels_beyond_contract"] = int((dmax > OUT_ATOL).sum())
    rep["worst_abs"] = rep["out_err_all"]
    # ---- contributor bookkeeping, as Gaussian ids
    st = hip_state(h, inp) if o["R"] > 0 else None
    if st is not None:
        hid = _last_and_median_ids(st["n_contrib"], st["ranges"], (st["entries"] & np.uint64(0xFFFFFFFF)).astype(np.uint32), W, H)
        oid = _last_and_median_ids(orc.state("n_contrib"), orc.state("ranges"), orc.state("point_list"), W, H)
        id_bad = (hid[0] != oid[0]) | (hid[1] != oid[1])
    else:
        id_bad = np.zeros(N, bool)
    rep["id_mismatch"] = int(id_bad.sum())
    rep["id_mismatch_unexplained"] = int((id_bad & ~suspect).sum())
    # flipped pixels: on a threshold AND visibly different
    flipped = suspect & ((dmax > OUT_ATOL_GUARD) | id_bad)
    rep["flipped_pixels"] = int(flipped.sum())
    rep["flipped_worst"] = float(dmax[flipped].max()) if flipped.any() else 0.0
    rep["flipped_max_margin"] = float(margins.min(axis=0)[flipped].max()) if flipped.any() else 0.0
    # A flipped pixel is not exempt either (verdict r4 item 4a: only their NUMBER used to be capped, not their size): the
    # oracle renders the pixel again with its near-threshold decisions -- a skip, a stop, a median move -- taken the other
    # way, in every combination (oracle.pixel_alternatives), and the HIP pixel must equal ONE of those renderings: all
    # ten maps within the guard bar and the same last / median contributor.  flipped_alt_err = the worst distance to the
    # best-matching alternative; flipped_unmatched = flipped pixels whose contributor ids match no alternative.
    rep["flipped_alt_err"], rep["flipped_unmatched"] = 0.0, 0
    if flipped.any():
        fidx = np.nonzero(flipped)[0]
        alt = oracle_mod.pixel_alternatives(orc, fidx, MARGIN)                      # [n, ALT, 12]
        hv = np.concatenate([h["color"].reshape(3, N), h["others"].reshape(7, N)], 0)[:, fidx].T.astype(np.float64)  # [n, 10]
        d = np.abs(alt[:, :, :10] - hv[:, None, :])
        if scale_aware:
            d = d / np.maximum(1.0, mag)[None, None, :]
        d = d.max(axis=2)                                                            # [n, ALT]
        if st is not None:
            ids_ok = (alt[:, :, 10] == hid[0][fidx][:, None]) & (alt[:, :, 11] == hid[1][fidx][:, None])
            rep["flipped_unmatched"] = int((~ids_ok.any(axis=1)).sum())
            d = np.where(ids_ok, d, np.inf)
        best = d.min(axis=1)
        rep["flipped_alt_err"] = float(best[np.isfinite(best)].max()) if np.isfinite(best).any() else 0.0
    # ---- gradients: a flipped pixel moves O(|cotangent|) between the Gaussians of its tile's list
    if "grads" in h and "grads" in o:
        P = inp["means3D"].shape[0]
        explained = np.zeros(P, bool)
        # A skip decision (alpha vs 1/255, depth vs near) within MARGIN of its threshold adds or removes ONE blended splat of
        # weight ~T/255 at that pixel: mostly invisible in the outputs (so the pixel is not among the `flipped` ones), but it
        # is that pixel's whole contribution to that Gaussian's gradient rows -- several per cent of a row that only a few
        # pixels feed.  Those Gaussians are explained too, and their pixels join the masked comparison below.
        # (exempt_skip_suspects=False -- round 4's gate, for runs on the reference's arithmetic: neither exempt nor masked)
        if exempt_skip_suspects:
            explained[sk_gid] = True
        rep["skip_suspect_pairs"] = int(len(sk_pix))
        if flipped.any():
            tiles_x = (W + 15) // 16
            fy, fx = np.divmod(np.nonzero(flipped)[0], W)
            olist, orng = orc.state("point_list"), orc.state("ranges")
            for t in np.unique((fy // 16) * tiles_x + fx // 16):
                explained[olist[orng[t, 0]:orng[t, 1]]] = True
        rep["explained_rows"] = int(explained.sum())
        g = {}
        for name in ("means3D", "scales", "rotations", "opacity", "sh", "colors", "transMat", "means2D"):
            a, b = h["grads"][name], o["grads"][name]
            if b.size == 0:
                assert a.size == 0, name
                continue
            assert a.shape == b.shape, name
            a, b = a.astype(np.float64).reshape(len(a), -1), b.astype(np.float64).reshape(len(b), -1)
            scale = np.abs(b).max() + 1e-30
            row_err = np.abs(a - b).max(axis=1)
            row_mag = np.abs(b).max(axis=1)
            un = ~explained
            big = un & (row_mag > ROW_FLOOR * scale)
            g[name] = dict(rel_unexplained=float(row_err[un].max() / scale) if un.any() else 0.0,
                           rel_all=float(row_err.max() / scale),
                           row_rel_unexplained=float((row_err[big] / row_mag[big]).max()) if big.any() else 0.0,
                           rows_beyond_contract=int((row_err > GRAD_RTOL * scale).sum()),
                           rows_beyond_contract_unexplained=int(((row_err > GRAD_RTOL * scale) & un).sum()),
                           l2=float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)),
                           l2_unexplained=float(np.linalg.norm((a - b)[un]) / (np.linalg.norm(b) + 1e-30)))
        rep["grads"] = g
        # ---- no row is exempt.  The gradient is a sum of independent per-pixel contributions, so with the cotangents
        # of the flipped pixels set to zero on BOTH sides every row -- the "explained" ones too -- must meet the same
        # bars: what remains unchecked is exactly the contribution of the (capped number of) flipped pixels themselves.
        rep["grads_masked"] = None
        masked = flipped.copy()
        if exempt_skip_suspects:
            masked[sk_pix] = True
        rep["masked_pixels"] = int(masked.sum())
        if masked_rerun and masked.any() and "cot" in h and "cot" in o:
            gc = np.array(h["cot"][0], np.float32, copy=True).reshape(3, N)
            go = np.array(h["cot"][1], np.float32, copy=True).reshape(7, N)
            gc[:, masked] = 0.0
            go[:, masked] = 0.0
            cot2 = (gc.reshape(3, H, W), go.reshape(7, H, W))
            hm = hip_backward_again(h, inp, cot2)
            om = orc.rasterize_gaussians_backward(cot2[0], cot2[1])
            gm = {}
            for name in g:
                a = hm[name].astype(np.float64).reshape(len(hm[name]), -1)
                b = om[name].astype(np.float64).reshape(len(om[name]), -1)
                scale = np.abs(b).max() + 1e-30
                row_err = np.abs(a - b).max(axis=1)
                row_mag = np.abs(b).max(axis=1)
                big = row_mag > ROW_FLOOR * scale
                gm[name] = dict(rel=float(row_err.max() / scale),
                                row_rel=float((row_err[big] / row_mag[big]).max()) if big.any() else 0.0,
                                l2=float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)),
                                # the rows the old gate exempted, now checked: their worst error with the flips masked
                                rel_formerly_explained=float(row_err[explained].max() / scale) if explained.any() else 0.0)
            rep["grads_masked"] = gm
    return rep


def assert_parity(h, o, inp, oracle_mod, tag="", out_atol=OUT_ATOL_GUARD, grad_rtol=GRAD_RTOL_GUARD,
                  row_rtol=ROW_RTOL_GUARD, max_flipped_frac=4e-5, scale_aware=False, masked_rerun=True,
                  exempt_skip_suspects=True):
    """The parity gate of the GPU tests: exact integers, guard bars on everything that is not explained by a
    decision threshold, and a cap on how much may be explained away.  `**STRICT` = round 4's gate (flip cap 2e-5 of the
    frame, Gaussians with a near-threshold skip decision neither exempt nor masked), which runs on the reference's
    arithmetic (option no_fastpath) still have to meet."""
    assert h["R"] == o["R"], tag
    np.testing.assert_array_equal(h["radii"], o["radii"], err_msg=tag)
    rep = parity_report(h, o, inp, oracle_mod, scale_aware=scale_aware, masked_rerun=masked_rerun,
                        exempt_skip_suspects=exempt_skip_suspects)
    N = rep["N"]
    if not scale_aware:
        # (the same allowance as for the flips below: every one of these pixels has to be a matched flip anyway)
        assert rep["pixels_beyond_contract"] <= max(2, CONTRACT_PIXELS_FRAC * N, 0.02 * rep["suspect_pixels"]), \
            (tag, "pixels beyond 1e-4 abs", rep)
    assert rep["out_err_unexplained"] <= out_atol, (tag, "output beyond the bar on a pixel that is on no threshold", rep)
    assert rep["id_mismatch_unexplained"] == 0, (tag, "contributor mismatch on a pixel that is on no threshold", rep)
    # how many pixels may flip: a share of the frame -- or, where the lists are so deep that a frame has thousands of
    # near-threshold decisions (tools/fuzz_sweep.py seed 90360: 12 instances per pixel, 3.3 % of the pixels within MARGIN, five of
    # them flipped at margins <= 1.2e-7), 2 % of the pixels that sit within MARGIN of a threshold: a HIP error of the order of
    # MARGIN itself would flip half of them.  (Every flipped pixel is checked against the oracle's alternative below.)
    assert rep["flipped_pixels"] <= max(2, max_flipped_frac * N, 0.02 * rep["suspect_pixels"]), (tag, "too many threshold flips", rep)
    assert rep["flipped_unmatched"] == 0, (tag, "a flipped pixel's contributors match none of the oracle's alternatives", rep)
    assert rep["flipped_alt_err"] <= out_atol, (tag, "a flipped pixel is not the oracle's pixel with the decision taken the other way", rep)
    for name, g in rep.get("grads", {}).items():
        assert g["rel_unexplained"] <= grad_rtol, (tag, name, g)
        assert g["row_rel_unexplained"] <= row_rtol, (tag, name, g)
        # relative L2 error of the whole tensor: over the rows no flip can have touched always; over all rows when
        # nothing flipped (a single flipped pixel moves O(|cotangent|) -- in a one-splat scene that is 0.1 % of everything)
        assert g["l2_unexplained"] <= GRAD_RTOL, (tag, name, g)
        assert rep["flipped_pixels"] > 0 or g["l2"] <= GRAD_RTOL, (tag, name, g)
    # with the flipped pixels' cotangents zeroed on both sides NO row is exempt (verdict r2 item 4): the rows that sit
    # in the tile list of a flip meet the same bars as everything else
    for name, g in (rep.get("grads_masked") or {}).items():
        assert g["rel"] <= grad_rtol, (tag, name, "masked", g)
        assert g["row_rel"] <= row_rtol, (tag, name, "masked", g)
        assert g["l2"] <= GRAD_RTOL, (tag, name, "masked", g)
    return rep
