"""Seeded synthetic scenes and pinhole cameras for tests and bench.py.

No dataset exists in the build container or on the GPU box, so every measurement and
parity test runs on these generators.  Camera matrices follow the reference conventions
(row-vector, `world_view_transform` = W2C^T, `full_proj_transform` = W2C^T @ Proj^T;
2dgs/scene/cameras.py:49-58, 2dgs/utils/graphics_utils.py:38-71) -- restated here and
pinned against the reference helpers by tests/golden/camera_*.npz.

Everything here is numpy on the host; callers move arrays to the device.
"""
import math
from typing import NamedTuple

import numpy as np

SH_C0 = 0.28209479177387814


class PinholeCamera(NamedTuple):
    """The attributes gaussian_renderer.render() reads from a camera
    (2dgs/gaussian_renderer/__init__.py:34-47)."""
    image_width: int
    image_height: int
    FoVx: float
    FoVy: float
    world_view_transform: np.ndarray  # [4,4] = W2C^T
    full_proj_transform: np.ndarray   # [4,4] = W2C^T @ Proj^T
    camera_center: np.ndarray         # [3]
    znear: float = 0.01
    zfar: float = 100.0

    @property
    def tanfovx(self):
        return math.tan(self.FoVx * 0.5)

    @property
    def tanfovy(self):
        return math.tan(self.FoVy * 0.5)


def world2view(R, t):
    """getWorld2View2(R, t) with translate=0, scale=1 (2dgs/utils/graphics_utils.py:38-49):
    Rt = [[R^T, t],[0,1]] as float32."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = np.asarray(R, dtype=np.float64).T
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    Rt = np.linalg.inv(C2W)
    return np.float32(Rt)


def projection_matrix(znear, zfar, fovX, fovY):
    """getProjectionMatrix (2dgs/utils/graphics_utils.py:51-71) in float32."""
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def make_camera(R, t, fovx, fovy, width, height, znear=0.01, zfar=100.0):
    """Camera.__init__ matrices (2dgs/scene/cameras.py:49-58).  R is the camera-to-world
    rotation (the reference stores R transposed, COLMAP style), t the world-to-camera translation."""
    wvt = world2view(R, t).T.copy()
    proj = projection_matrix(znear, zfar, fovx, fovy).T.copy()
    full = (wvt.astype(np.float32) @ proj.astype(np.float32)).astype(np.float32)
    center = np.linalg.inv(wvt.astype(np.float64))[3, :3].astype(np.float32)
    return PinholeCamera(int(width), int(height), float(fovx), float(fovy), np.ascontiguousarray(wvt, np.float32),
                         np.ascontiguousarray(full, np.float32), center, znear, zfar)


def look_at_camera(eye, target, up, fovx, width, height):
    """Pinhole camera at `eye` looking at `target` (+z forward, +x right, +y down)."""
    eye = np.asarray(eye, np.float64)
    fwd = np.asarray(target, np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.asarray(up, np.float64))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    c2w_R = np.stack([right, down, fwd], axis=1)  # columns = camera axes in world
    t = -c2w_R.T @ eye
    fovy = 2.0 * math.atan(math.tan(fovx / 2) * height / width)
    return make_camera(c2w_R, t, fovx, fovy, width, height)


def rgb2sh(rgb):
    return (rgb - 0.5) / SH_C0


class Scene(NamedTuple):
    means3D: np.ndarray    # [P,3]
    scales: np.ndarray     # [P,2] post-activation
    rotations: np.ndarray  # [P,4] (w,x,y,z) normalised
    opacities: np.ndarray  # [P,1] post-sigmoid
    shs: np.ndarray        # [P,16,3]


def scene_random(P, seed=0, fov_deg=60.0, width=256, height=256, opacity_max=1.0):
    """S1 (= BASELINE config 1): random Gaussians in front of an identity camera, 2 % of
    them behind/near the near plane for cull coverage."""
    rng = np.random.default_rng(seed)
    z = rng.uniform(0.5, 6.0, P)
    ncull = max(1, P // 50) if P >= 50 else 0
    if ncull:
        z[rng.choice(P, ncull, replace=False)] = rng.uniform(-1.0, 0.2, ncull)
    half = math.tan(math.radians(fov_deg) / 2) * 1.3
    xy = rng.uniform(-half, half, (P, 2)) * np.abs(z)[:, None]
    means = np.concatenate([xy, z[:, None]], 1).astype(np.float32)
    scales = np.exp(rng.uniform(math.log(0.005), math.log(0.15), (P, 2))).astype(np.float32)
    rot = rng.normal(size=(P, 4))
    rot = (rot / np.linalg.norm(rot, axis=1, keepdims=True)).astype(np.float32)
    opac = (1.0 / (1.0 + np.exp(-rng.normal(0, 1.5, (P, 1))))).astype(np.float32)
    opac = np.minimum(opac, np.float32(opacity_max))
    shs = np.zeros((P, 16, 3), np.float32)
    shs[:, 0, :] = rgb2sh(rng.uniform(0, 1, (P, 3)))
    shs[:, 1:, :] = rng.normal(0, 0.1, (P, 15, 3))
    cam = make_camera(np.eye(3), np.zeros(3), math.radians(fov_deg),
                      2.0 * math.atan(math.tan(math.radians(fov_deg) / 2) * height / width), width, height)
    return Scene(means, scales, rot, opac, shs), cam


def _face_frames():
    # (origin, u axis, v axis, normal) of the six faces of a box centred at 0
    return [((0, 0, 1), (1, 0, 0), (0, 1, 0)), ((0, 0, -1), (1, 0, 0), (0, 1, 0)),
            ((0, 1, 0), (1, 0, 0), (0, 0, 1)), ((0, -1, 0), (1, 0, 0), (0, 0, 1)),
            ((1, 0, 0), (0, 1, 0), (0, 0, 1)), ((-1, 0, 0), (0, 1, 0), (0, 0, 1))]


def _rotmat_to_quat(Rm):
    """Rotation matrices [n,3,3] -> quaternions (w,x,y,z); vectorised Shepperd branches."""
    m = Rm
    n = m.shape[0]
    q = np.zeros((n, 4))
    tr = m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2]
    c0 = tr > 0
    c1 = ~c0 & (m[:, 0, 0] > m[:, 1, 1]) & (m[:, 0, 0] > m[:, 2, 2])
    c2 = ~c0 & ~c1 & (m[:, 1, 1] > m[:, 2, 2])
    c3 = ~c0 & ~c1 & ~c2
    with np.errstate(invalid="ignore"):
        s = np.sqrt(np.maximum(tr + 1.0, 1e-30)) * 2
        q[c0] = np.stack([0.25 * s, (m[:, 2, 1] - m[:, 1, 2]) / s, (m[:, 0, 2] - m[:, 2, 0]) / s,
                          (m[:, 1, 0] - m[:, 0, 1]) / s], 1)[c0]
        s = np.sqrt(np.maximum(1.0 + m[:, 0, 0] - m[:, 1, 1] - m[:, 2, 2], 1e-30)) * 2
        q[c1] = np.stack([(m[:, 2, 1] - m[:, 1, 2]) / s, 0.25 * s, (m[:, 0, 1] + m[:, 1, 0]) / s,
                          (m[:, 0, 2] + m[:, 2, 0]) / s], 1)[c1]
        s = np.sqrt(np.maximum(1.0 + m[:, 1, 1] - m[:, 0, 0] - m[:, 2, 2], 1e-30)) * 2
        q[c2] = np.stack([(m[:, 0, 2] - m[:, 2, 0]) / s, (m[:, 0, 1] + m[:, 1, 0]) / s, 0.25 * s,
                          (m[:, 1, 2] + m[:, 2, 1]) / s], 1)[c2]
        s = np.sqrt(np.maximum(1.0 + m[:, 2, 2] - m[:, 0, 0] - m[:, 1, 1], 1e-30)) * 2
        q[c3] = np.stack([(m[:, 1, 0] - m[:, 0, 1]) / s, (m[:, 0, 2] + m[:, 2, 0]) / s, (m[:, 1, 2] + m[:, 2, 1]) / s,
                          0.25 * s], 1)[c3]
    return q


def scene_room(P, seed=0, size=(6.0, 4.0, 3.0), scale_mean=0.02, scale_sigma=0.4, jitter_deg=5.0):
    """S2/S3/S5 stand-in for an indoor scan (Replica room0 / ScanNet++): P surfels on the six
    faces of a `size` box, normals = face normals +- jitter."""
    rng = np.random.default_rng(seed)
    sx, sy, sz = size
    half = np.array([sx, sy, sz]) / 2
    faces = _face_frames()
    areas = np.array([sx * sy, sx * sy, sx * sz, sx * sz, sy * sz, sy * sz])
    fid = rng.choice(6, P, p=areas / areas.sum())
    uv = rng.uniform(-1, 1, (P, 2))
    means = np.zeros((P, 3))
    quats = np.zeros((P, 4))
    # per-face rotation batches (vectorised; a python loop over P would take minutes at 3M)
    jit = np.radians(jitter_deg)
    for f, (nrm, ua, va) in enumerate(faces):
        m = np.nonzero(fid == f)[0]
        if m.size == 0:
            continue
        nrm, ua, va = np.array(nrm, float), np.array(ua, float), np.array(va, float)
        means[m] = nrm * half + uv[m, :1] * ua * half + uv[m, 1:] * va * half
        # small random rotation about a random in-plane axis + spin about the normal
        spin = rng.uniform(0, 2 * math.pi, m.size)
        tilt = rng.normal(0, jit, m.size)
        tax = rng.uniform(0, 2 * math.pi, m.size)
        base = np.stack([ua, va, -nrm * np.sign(np.dot(np.cross(ua, va), nrm))], 1)  # columns u,v,n(inward/outward)
        cs, sn = np.cos(spin), np.sin(spin)
        Rspin = np.zeros((m.size, 3, 3)); Rspin[:, 0, 0] = cs; Rspin[:, 0, 1] = -sn; Rspin[:, 1, 0] = sn; Rspin[:, 1, 1] = cs; Rspin[:, 2, 2] = 1
        ax = np.stack([np.cos(tax), np.sin(tax), np.zeros_like(tax)], 1)
        K = np.zeros((m.size, 3, 3))
        K[:, 0, 1] = -ax[:, 2]; K[:, 0, 2] = ax[:, 1]; K[:, 1, 0] = ax[:, 2]; K[:, 1, 2] = -ax[:, 0]; K[:, 2, 0] = -ax[:, 1]; K[:, 2, 1] = ax[:, 0]
        Rtilt = np.eye(3)[None] + np.sin(tilt)[:, None, None] * K + (1 - np.cos(tilt))[:, None, None] * (K @ K)
        Rm = base[None] @ Rtilt @ Rspin
        # make proper rotations (det +1)
        det = np.linalg.det(Rm)
        Rm[det < 0, :, 2] *= -1
        quats[m] = _rotmat_to_quat(Rm)
    scales = np.exp(rng.normal(math.log(scale_mean), scale_sigma, (P, 2))).astype(np.float32)
    opac = (1.0 / (1.0 + np.exp(-rng.normal(1.0, 1.5, (P, 1))))).astype(np.float32)
    shs = np.zeros((P, 16, 3), np.float32)
    shs[:, 0, :] = rgb2sh(rng.uniform(0, 1, (P, 3)))
    shs[:, 1:, :] = rng.normal(0, 0.05, (P, 15, 3))
    quats = (quats / np.linalg.norm(quats, axis=1, keepdims=True)).astype(np.float32)
    return Scene(means.astype(np.float32), scales, quats, opac, shs)


def room_cameras(n, width, height, fovx_deg=90.0, radius=1.2, height_y=0.0, seed=0):
    """`n` cameras on an interior ring looking outward-ish across the room."""
    cams = []
    for i in range(n):
        a = 2 * math.pi * i / max(n, 1) + 0.1 * seed
        eye = (radius * math.cos(a), height_y, radius * math.sin(a) * 0.6)
        target = (-2.5 * math.cos(a), 0.2 * math.sin(3 * a), -1.5 * math.sin(a))
        cams.append(look_at_camera(eye, target, (0, 1, 0), math.radians(fovx_deg), width, height))
    return cams
