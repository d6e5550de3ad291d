"""PLY scene format of the reference (2dgs/scene/gaussian_model.py:276-315 save_ply, :441-493 load_ply):
header text, attribute order, channel-major SH layout, round trips, and foreign-but-valid encodings."""
import numpy as np
import pytest
import torch

from g4splat_amd import ply_io
from g4splat_amd.gaussian_model import GaussianModel


def _model(P=37, D=3, mip=False, seed=0):
    rng = np.random.default_rng(seed)
    m = GaussianModel(sh_degree=D, use_mip_filter=mip)
    m.create_from_parameters(torch.tensor(rng.normal(size=(P, 3)).astype(np.float32)),
                             torch.tensor(rng.uniform(0.01, 0.3, (P, 2)).astype(np.float32)),
                             torch.tensor(rng.normal(size=(P, 4)).astype(np.float32)),
                             torch.tensor(rng.uniform(0, 1, (P, 3)).astype(np.float32)))
    with torch.no_grad():
        m._features_rest += torch.tensor(rng.normal(size=tuple(m._features_rest.shape)).astype(np.float32))
        m._opacity += torch.tensor(rng.normal(size=(P, 1)).astype(np.float32))
    if mip:
        m.mip_filter = torch.tensor(rng.uniform(0, 0.01, (P, 1)).astype(np.float32))
    return m


def test_header_and_layout_match_the_reference_writer(tmp_path):
    m = _model()
    path = str(tmp_path / "sub" / "point_cloud.ply")
    m.save_ply(path)  # creates the directory like mkdir_p
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().split("\n")
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    names = [l.split()[-1] for l in lines[3:] if l]
    assert all(l.startswith("property float ") for l in lines[3:] if l)
    want = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)] + \
           ["opacity", "scale_0", "scale_1", "rot_0", "rot_1", "rot_2", "rot_3"]
    assert names == want  # construct_list_of_attributes, :276-291
    tab = np.frombuffer(body, "<f4").reshape(37, len(want))
    np.testing.assert_array_equal(tab[:, 0:3], m._xyz.detach().numpy())
    assert not tab[:, 3:6].any()  # normals are zeros (:298)
    # channel-major SH: f_rest_k = features_rest[:, k % 15, k // 15]   (transpose(1,2).flatten, :300)
    fr = m._features_rest.detach().numpy()
    for k in (0, 1, 14, 15, 31, 44):
        np.testing.assert_array_equal(tab[:, 9 + k], fr[:, k % 15, k // 15])
    np.testing.assert_array_equal(tab[:, 6:9], m._features_dc.detach().numpy()[:, 0, :])
    np.testing.assert_array_equal(tab[:, 54], m._opacity.detach().numpy()[:, 0])
    np.testing.assert_array_equal(tab[:, 57:61], m._rotation.detach().numpy())


@pytest.mark.parametrize("D,mip", [(3, False), (3, True), (1, False), (0, False)])
def test_round_trip(tmp_path, D, mip):
    m = _model(P=50, D=D, mip=mip, seed=D)
    path = str(tmp_path / "pc.ply")
    m.save_ply(path)
    n = GaussianModel(sh_degree=D)
    n.load_ply(path, device="cpu")
    for a, b in zip(m.parameters(), n.parameters()):
        assert a.shape == b.shape and torch.equal(a.detach(), b.detach())
        assert b.requires_grad
    assert n.use_mip_filter == mip and n.active_sh_degree == D
    if mip:
        assert torch.equal(n.mip_filter, m.mip_filter)
    with pytest.raises(ValueError):
        GaussianModel(sh_degree=D + 1).load_ply(path, device="cpu")  # the reference asserts the f_rest count (:466)


def test_reader_accepts_other_valid_encodings(tmp_path):
    """What plyfile may legally emit for the same data: big endian, ascii, doubles, shuffled property order."""
    m = _model(P=9, D=1)
    path = str(tmp_path / "a.ply")
    m.save_ply(path)
    ref = ply_io.read_gaussian_ply(path, 1)
    v = ply_io.read_ply_vertices(path)
    names = list(v)[::-1]  # reversed order
    P = 9
    big = str(tmp_path / "big.ply")
    with open(big, "wb") as f:
        f.write(("ply\nformat binary_big_endian 1.0\ncomment made by a test\nelement vertex 9\n" +
                 "".join(f"property double {n}\n" for n in names) + "element face 0\nproperty list uchar int vertex_indices\nend_header\n").encode())
        f.write(np.stack([np.asarray(v[n], np.float64) for n in names], 1).astype(">f8").tobytes())
    asc = str(tmp_path / "asc.ply")
    with open(asc, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 9\n" + "".join(f"property float {n}\n" for n in names) + "end_header\n")
        for i in range(P):
            f.write(" ".join(repr(float(v[n][i])) for n in names) + "\n")
    for other in (big, asc):
        got = ply_io.read_gaussian_ply(other, 1)
        for k in ref:
            if ref[k] is None:
                assert got[k] is None
            else:
                np.testing.assert_array_equal(got[k], ref[k])


def test_empty_scene(tmp_path):
    m = _model(P=0)
    path = str(tmp_path / "e.ply")
    m.save_ply(path)
    n = GaussianModel(sh_degree=3)
    n.load_ply(path, device="cpu")
    assert n._xyz.shape == (0, 3) and n._features_rest.shape == (0, 15, 3)


def test_reads_a_scene_file_built_byte_by_byte_in_the_reference_layout():
    """tests/golden/scene_ref_layout.ply was packed with struct.pack, attribute by attribute, from the layout of
    gaussian_model.py:276-315 (tests/golden/make_golden_ply.py shares no code with ply_io.py): the reader must return
    exactly the arrays it was built from, and GaussianModel.load_ply the model the reference's load_ply (:441-493) builds."""
    import os
    import torch
    from g4splat_amd.gaussian_model import GaussianModel
    from g4splat_amd.ply_io import read_gaussian_ply
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    want = np.load(os.path.join(gold, "scene_ref_layout.npz"))
    got = read_gaussian_ply(os.path.join(gold, "scene_ref_layout.ply"), 3)
    for k in want.files:
        assert got[k].dtype == np.float32 and got[k].shape == want[k].shape, k
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    m = GaussianModel(3)
    m.load_ply(os.path.join(gold, "scene_ref_layout.ply"), device="cpu")
    assert m.use_mip_filter and m.active_sh_degree == 3
    assert torch.equal(m._features_rest.detach(), torch.tensor(want["features_rest"]))
    assert m._features_dc.shape == (200, 1, 3) and m._features_rest.shape == (200, 15, 3)
    # the activations the renderer sees (mip filter on): scaling^2 + filter^2, opacity rescaled
    s = np.exp(want["scaling"])
    np.testing.assert_allclose(m.get_scaling.detach().numpy(), np.sqrt(s * s + want["mip_filter"] ** 2), rtol=1e-6)
