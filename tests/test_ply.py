"""Row f3: the Gaussian PLY format of `gs-export gaussian-splat`
(gs_toolkit/scripts/exporter.py:88-147): header text, property order, channel-major
f_rest, raw values, byte-exact round trip."""
import numpy as np
import pytest

from gs_io import read_gaussian_ply, write_gaussian_ply
from harness.train import blob_scene


@pytest.mark.parametrize("deg,n", [(3, 257), (0, 5), (1, 1)])
def test_roundtrip_and_layout(tmp_path, deg, n):
    raw = blob_scene(n, seed=3, sh_degree=deg)
    path = str(tmp_path / "gaussians.ply")
    write_gaussian_ply(path, raw)
    blob = open(path, "rb").read()
    K1 = (deg + 1) ** 2 - 1
    names = (["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(3 * K1)]
             + ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"])
    header = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n
              + "".join(f"property float {a}\n" for a in names) + "end_header\n").encode()
    assert blob.startswith(header)
    body = np.frombuffer(blob[len(header):], "<f4").reshape(n, len(names))
    assert body.shape[1] == 17 + 3 * K1
    assert np.array_equal(body[:, 0:3], raw["means"]) and np.all(body[:, 3:6] == 0)
    assert np.array_equal(body[:, 6:9], raw["features_dc"])
    if K1:
        # channel-major: f_rest_{c*K1 + k} = features_rest[n, k, c]
        for c in range(3):
            for k in (0, K1 - 1):
                assert np.array_equal(body[:, 9 + c * K1 + k], raw["features_rest"][:, k, c])
    o = 9 + 3 * K1
    assert np.array_equal(body[:, o], raw["opacities"][:, 0])          # logits, not sigmoid
    assert np.array_equal(body[:, o + 1:o + 4], raw["scales"])         # log-scales
    assert np.array_equal(body[:, o + 4:o + 8], raw["quats"])          # un-normalised wxyz
    back = read_gaussian_ply(path)
    for k, v in raw.items():
        assert back[k].shape == v.shape and np.array_equal(back[k], v), k


def test_reader_rejects_other_files(tmp_path):
    p = tmp_path / "x.ply"
    p.write_bytes(b"ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nend_header\n0\n")
    with pytest.raises(ValueError):
        read_gaussian_ply(str(p))
    p.write_bytes(b"plx\n")
    with pytest.raises(ValueError):
        read_gaussian_ply(str(p))


@pytest.mark.parametrize("K", [1, 4, 16])
def test_empty_model_keeps_its_sh_degree(tmp_path, K):
    """N = 0 (everything culled): the file still lists the f_rest properties of the model's SH degree."""
    f = np.float32
    raw = {"means": np.zeros((0, 3), f), "scales": np.zeros((0, 3), f), "quats": np.zeros((0, 4), f),
           "opacities": np.zeros((0, 1), f), "features_dc": np.zeros((0, 3), f), "features_rest": np.zeros((0, K - 1, 3), f)}
    path = str(tmp_path / "empty.ply")
    write_gaussian_ply(path, raw)
    back = read_gaussian_ply(path)
    for k, v in raw.items():
        assert back[k].shape == v.shape, k
