"""On-disk format of a trained scene: the PLY that
2dgs/scene/gaussian_model.py:276-315 (`construct_list_of_attributes`, `save_ply`) writes with `plyfile`
and `load_ply` (:441-493) reads back, restated on numpy alone (plyfile is not a dependency here).

Layout: one `vertex` element, every property `float` (f4), binary little endian, in this order:
    x y z  nx ny nz  f_dc_0..2  f_rest_0..(3*((D+1)^2-1)-1)  opacity  scale_0..1  rot_0..3  [mip_filter]
`f_dc` / `f_rest` are stored CHANNEL-major (`transpose(1,2).flatten`), i.e. f_rest_k = features_rest[:, k % n, k // n]
with n = (D+1)^2 - 1; normals are zeros; values are the raw (pre-activation) parameters.
The reader accepts what plyfile can produce for such a file: binary little/big endian or ascii, properties of
any scalar type, in any order."""
import os

import numpy as np

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4",
              "float32": "f4", "double": "f8", "float64": "f8"}


def attribute_names(n_rest_coeffs, n_scale=2, n_rot=4, mip_filter=False):
    """gaussian_model.py:276-291."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(3)]
    names += [f"f_rest_{i}" for i in range(3 * n_rest_coeffs)]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(n_scale)]
    names += [f"rot_{i}" for i in range(n_rot)]
    if mip_filter:
        names.append("mip_filter")
    return names


def write_gaussian_ply(path, xyz, features_dc, features_rest, opacity, scaling, rotation, mip_filter=None):
    """Arrays as the model holds them: xyz [P,3], features_dc [P,1,3], features_rest [P,n,3], opacity [P,1],
    scaling [P,2], rotation [P,4], optional mip_filter [P,1] (gaussian_model.py:293-315)."""
    a = lambda t: np.ascontiguousarray(np.asarray(t, dtype=np.float32))
    xyz, fdc, frest = a(xyz), a(features_dc), a(features_rest)
    P = xyz.shape[0]
    flat = lambda t: a(t).reshape(P, int(np.prod(np.shape(t)[1:])))  # (also for P == 0, where -1 is ambiguous)
    cols = [xyz, np.zeros_like(xyz), flat(fdc.transpose(0, 2, 1)), flat(frest.transpose(0, 2, 1)), flat(opacity),
            flat(scaling), flat(rotation)]
    if mip_filter is not None:
        cols.append(flat(mip_filter))
    table = np.concatenate(cols, axis=1).astype("<f4")
    names = attribute_names(frest.shape[1], cols[5].shape[1], cols[6].shape[1], mip_filter is not None)
    assert table.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {P}\n" + \
             "".join(f"property float {n}\n" for n in names) + "end_header\n"
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)  # mkdir_p, gaussian_model.py:294
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(table.tobytes())


def read_ply_vertices(path):
    """-> dict name -> float64/np array [P] of the first element of the file (must be `vertex`-like scalars only)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_first = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if count is None:
                    count, in_first = int(tok[2]), True
                else:
                    in_first = False
            elif tok[0] == "property" and in_first:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not supported in the first element")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt is None or count is None:
            raise ValueError(f"{path}: incomplete PLY header")
        if fmt == "ascii":
            rows = np.loadtxt(f, max_rows=count, ndmin=2) if count else np.zeros((0, len(props)))
            return {n: rows[:, i] for i, (n, _t) in enumerate(props)}
        end = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, end + t) for n, t in props])
        data = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
    return {n: data[n] for n, _t in props}


def read_gaussian_ply(path, max_sh_degree=3):
    """gaussian_model.py:441-493 -> dict of float32 arrays shaped like the model's parameters."""
    v = read_ply_vertices(path)
    P = len(v["x"])
    col = lambda names: np.stack([np.asarray(v[n], np.float32) for n in names], axis=1) if names else np.zeros((P, 0), np.float32)
    by_index = lambda prefix: sorted((n for n in v if n.startswith(prefix)), key=lambda n: int(n.split("_")[-1]))
    rest = by_index("f_rest_")
    n_rest = (max_sh_degree + 1) ** 2 - 1
    if len(rest) != 3 * n_rest:  # the reference asserts the same (:466)
        raise ValueError(f"{path}: {len(rest)} f_rest properties, expected {3 * n_rest} for SH degree {max_sh_degree}")
    out = {
        "xyz": col(["x", "y", "z"]),
        "features_dc": col(["f_dc_0", "f_dc_1", "f_dc_2"]).reshape(P, 3, 1).transpose(0, 2, 1).copy(),
        "features_rest": col(rest).reshape(P, 3, n_rest).transpose(0, 2, 1).copy(),
        "opacity": col(["opacity"]),
        "scaling": col(by_index("scale_")),
        "rotation": col(by_index("rot")),
        "mip_filter": col(["mip_filter"]) if "mip_filter" in v else None,
    }
    return out
