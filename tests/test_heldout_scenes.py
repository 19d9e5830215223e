"""CPU: the held-out scene families of harness.scene.make_heldout_scene are what their docstring says --
shapes and dtypes of `make_scene`, finite values, and the property each family exists for, checked through the oracle's
projection: a heavy spatially correlated tail (room), screen-filling faint splats at the head of every list (floaters),
axis ratios of 1:20 and more (needles)."""
import numpy as np
import pytest

from harness import scene as S
from oracle import oracle as O


@pytest.mark.parametrize("kind", S.HELDOUT_KINDS)
def test_heldout_scene_families(kind):
    W, H, n = 480, 270, 20_000
    cam = S.make_camera(W, H)
    sc = S.make_heldout_scene(kind, n, cam, sh_degree=3)
    ref = S.make_scene(n, cam, sh_degree=3)
    assert set(sc) == set(ref)
    for k in sc:
        assert sc[k].shape == ref[k].shape and sc[k].dtype == np.float32 and np.isfinite(sc[k]).all(), k
    assert (sc["scales"] > 0).all() and (sc["opacities"] > 0).all() and (sc["opacities"] < 1).all()
    assert np.allclose(np.linalg.norm(sc["quats"], axis=-1), 1.0, atol=1e-5)
    again = S.make_heldout_scene(kind, n, cam, sh_degree=3)
    assert all(np.array_equal(sc[k], again[k]) for k in sc)  # seeded: CPU oracle and GPU see the same inputs
    cov3d, xys, depths, radii, conics, comp, tiles = O.project_gaussians_forward(
        n, sc["means3d"], sc["scales"], 1.0, sc["quats"], cam.viewmat[:3], cam.projmat, cam.fx, cam.fy, cam.cx, cam.cy,
        H, W, 16, 0.01)
    visible = radii > 0
    assert visible.mean() > 0.8
    ratio = sc["scales"].max(axis=1) / sc["scales"].min(axis=1)
    if kind == "needles":
        assert ratio.min() >= 19.9 and np.median(tiles[visible]) > 10      # long thin footprints over many tiles
    elif kind == "floaters":
        near = np.zeros(n, bool)
        near[:300] = True                                                  # (the generator puts its 300 floaters first)
        assert depths[near & visible].max() < 1.0 and sc["opacities"][near].max() <= 0.25
        assert np.median(tiles[near & visible]) >= 50                      # each covers a large part of the 510-tile grid
        assert np.median(tiles[~near & visible]) <= 4
    else:  # room
        walls = sc["scales"].max(axis=1) > 0.2
        assert 300 <= walls.sum() <= 4000 and sc["opacities"][walls].min() >= 0.85
        assert depths[walls & visible].min() > depths[~walls & visible].max() * 0.4  # the walls lie behind most of the detail
        assert np.median(tiles[walls & visible]) > 8 * np.median(tiles[~walls & visible])


def test_unknown_family_raises():
    with pytest.raises(ValueError):
        S.make_heldout_scene("garden", 10, S.make_camera(64, 48))
