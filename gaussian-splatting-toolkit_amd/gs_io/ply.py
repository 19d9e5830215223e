"""Gaussian-splat PLY files, as `gs-export gaussian-splat` writes them.

Layout restated from `ExportGaussianSplat.main` / `construct_list_of_attributes`
(gs_toolkit/scripts/exporter.py:88-147): one `vertex` element, every property
`float` (f4), binary little-endian (what plyfile's `PlyData([el]).write` emits by
default), in this order:

    x y z  nx ny nz  f_dc_0..2  f_rest_0..(3*(K-1)-1)  opacity  scale_0..2  rot_0..3

with RAW (pre-activation) values: log-scales, logit opacities, un-normalised
(w,x,y,z) quaternions, normals all zero, and `f_rest` stored CHANNEL-major
(`features_rest.transpose(1, 2).flatten(1)`: f_rest_{c*(K-1)+k} = features_rest[n,k,c]).
No dependency on plyfile (not installed here); numpy only.
"""
from typing import Dict

import numpy as np


def _attribute_names(n_rest: int):
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    names += [f"f_rest_{i}" for i in range(n_rest)]
    names += ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    return names


def write_gaussian_ply(path: str, params: Dict[str, np.ndarray]) -> None:
    """`params`: means [N,3], features_dc [N,3], features_rest [N,K-1,3], opacities [N,1],
    scales [N,3], quats [N,4] (raw values, as held by the model's `gauss_params`)."""
    xyz = np.asarray(params["means"], np.float32)
    n = xyz.shape[0]
    f_dc = np.asarray(params["features_dc"], np.float32).reshape(n, 3)
    rest = np.asarray(params["features_rest"], np.float32)
    # channel-major; the width is spelled out so that an empty model (N = 0) keeps its SH degree
    f_rest = np.ascontiguousarray(rest.transpose(0, 2, 1)).reshape(n, rest.shape[1] * rest.shape[2])
    cols = [xyz, np.zeros_like(xyz), f_dc, f_rest, np.asarray(params["opacities"], np.float32).reshape(n, 1),
            np.asarray(params["scales"], np.float32).reshape(n, 3),
            np.asarray(params["quats"], np.float32).reshape(n, 4)]
    table = np.ascontiguousarray(np.concatenate(cols, axis=1).astype("<f4"))
    names = _attribute_names(f_rest.shape[1])
    assert table.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n
    header += "".join(f"property float {a}\n" for a in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(table.tobytes())


def read_gaussian_ply(path: str) -> Dict[str, np.ndarray]:
    """Inverse of `write_gaussian_ply`; also accepts files written by the toolkit
    itself or by the original 3DGS code (same property names)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, n, props, in_vertex = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError("unexpected end of PLY header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] not in ("float", "float32"):
                    raise ValueError(f"unsupported property type {tok[1]}")
                props.append(tok[2])
            elif tok[0] == "end_header":
                break
        if fmt != "binary_little_endian" or n is None:
            raise ValueError("only binary_little_endian PLY files with a vertex element are supported")
        data = np.frombuffer(f.read(n * len(props) * 4), dtype="<f4").reshape(n, len(props))
    col = {p: i for i, p in enumerate(props)}
    take = lambda names: np.ascontiguousarray(data[:, [col[a] for a in names]], dtype=np.float32)
    n_rest = sum(1 for p in props if p.startswith("f_rest_"))
    if n_rest % 3:
        raise ValueError("f_rest count must be a multiple of 3")
    rest = take([f"f_rest_{i}" for i in range(n_rest)]).reshape(n, 3, n_rest // 3).transpose(0, 2, 1)
    return {
        "means": take(["x", "y", "z"]),
        "features_dc": take(["f_dc_0", "f_dc_1", "f_dc_2"]),
        "features_rest": np.ascontiguousarray(rest),
        "opacities": take(["opacity"]),
        "scales": take(["scale_0", "scale_1", "scale_2"]),
        "quats": take(["rot_0", "rot_1", "rot_2", "rot_3"]),
    }
