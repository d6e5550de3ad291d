"""A scene file in the reference's on-disk format, built BYTE BY BYTE -- independently of g4splat_amd/ply_io.py.

    scene_ref_layout.ply   what `GaussianModel.save_ply` (2d-gaussian-splatting/scene/gaussian_model.py:293-315) writes
                           through plyfile for a 200-surfel model with SH degree 3 and a mip filter: an ASCII header
                           ("ply / format binary_little_endian 1.0 / element vertex N / property float <name> ... /
                           end_header", names in construct_list_of_attributes order, :276-291) followed by one
                           little-endian float32 row per Gaussian
                               x y z  nx ny nz (zeros)  f_dc_0..2  f_rest_0..44  opacity  scale_0..1  rot_0..3  mip_filter
                           with the SH blocks CHANNEL-major (`transpose(1, 2).flatten(start_dim=1)`, :299-300):
                           f_rest_{c * 15 + k} = features_rest[p, k, c].
    scene_ref_layout.npz   the same parameters as arrays shaped like the model's tensors (what load_ply must return).

Every row is packed with struct.pack from python scalars, attribute by attribute, in loops that spell the layout out --
no numpy transposes, no code shared with the reader under test.  (plyfile itself is not installed in this image, so
the reference's writer cannot be run; its format is restated from the lines cited above.)

    python tests/golden/make_golden_ply.py"""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main():
    from g4splat_amd import synthetic
    P, n_rest = 200, 15
    scene = synthetic.scene_room(P, seed=5, size=(2.0, 1.5, 1.2), scale_mean=0.05)
    rng = np.random.default_rng(6)
    xyz = scene.means3D.astype(np.float32)
    f_dc = scene.shs[:, 0:1, :].astype(np.float32)                    # [P,1,3]
    f_rest = rng.normal(0, 0.08, (P, n_rest, 3)).astype(np.float32)   # [P,15,3]
    opacity = rng.normal(1.0, 1.5, (P, 1)).astype(np.float32)         # raw (pre-sigmoid)
    scaling = np.log(scene.scales).astype(np.float32)                 # raw (pre-exp)
    rotation = (scene.rotations * rng.uniform(0.5, 2.0, (P, 1))).astype(np.float32)  # raw (un-normalised)
    mip = rng.uniform(0.001, 0.01, (P, 1)).astype(np.float32)

    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += ["f_dc_%d" % i for i in range(3)]
    names += ["f_rest_%d" % i for i in range(3 * n_rest)]
    names += ["opacity", "scale_0", "scale_1", "rot_0", "rot_1", "rot_2", "rot_3", "mip_filter"]
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % P
    for n in names:
        header += "property float %s\n" % n
    header += "end_header\n"
    body = bytearray()
    for p in range(P):
        row = [float(xyz[p, 0]), float(xyz[p, 1]), float(xyz[p, 2]), 0.0, 0.0, 0.0]
        for c in range(3):                      # f_dc: channel-major, one coefficient
            row.append(float(f_dc[p, 0, c]))
        for c in range(3):                      # f_rest: all 15 coefficients of channel 0, then channel 1, then 2
            for k in range(n_rest):
                row.append(float(f_rest[p, k, c]))
        row.append(float(opacity[p, 0]))
        row += [float(scaling[p, 0]), float(scaling[p, 1])]
        row += [float(rotation[p, i]) for i in range(4)]
        row.append(float(mip[p, 0]))
        assert len(row) == len(names) == 62
        body += struct.pack("<62f", *row)
    with open(os.path.join(HERE, "scene_ref_layout.ply"), "wb") as f:
        f.write(header.encode("ascii"))
        f.write(bytes(body))
    np.savez(os.path.join(HERE, "scene_ref_layout.npz"), xyz=xyz, features_dc=f_dc, features_rest=f_rest, opacity=opacity,
             scaling=scaling, rotation=rotation, mip_filter=mip)
    print("wrote scene_ref_layout.ply", len(header) + len(body), "bytes")


if __name__ == "__main__":
    main()
