"""CPU: the host logic of the reference's training schedule as harness.train restates it (vanilla_gs.py:646-669
coarse-to-fine resolution, :688-690 random background, :859-881 ground truth downscaled and composited)."""
import math

import numpy as np
import torch

from harness import scene as S
from harness.train import (composite_with_background, downscale_factor, downscale_image, orbit_cameras,
                           rescale_camera)


def test_downscale_factor_follows_the_reference_schedule():
    # num_downscales 2, resolution_schedule 2000: 4 until step 1999, 2 until 3999, then 1 (vanilla_gs.py:646-657)
    assert [downscale_factor(s, 2, 2000) for s in (0, 1999, 2000, 3999, 4000, 6999, 30000)] == [4, 4, 2, 2, 1, 1, 1]
    assert downscale_factor(0, 0, 2000) == 1 and downscale_factor(123, 3, 100) == 4


def test_rescaled_camera_is_what_rescale_output_resolution_and_get_outputs_build():
    cam = orbit_cameras(4, 1920, 1080, radius=5.0)[1]
    for d in (2, 4):
        c = rescale_camera(cam, d)
        assert (c.width, c.height) == (1920 // d, 1080 // d)
        assert c.fx == cam.fx / d and c.cx == cam.cx / d and c.fy == cam.fy / d and c.cy == cam.cy / d
        # the fields of view come from the NEW width / fx: unchanged when the size divides (cameras.py:1208-1213)
        fovx, fovy = 2 * math.atan(c.width / (2 * c.fx)), 2 * math.atan(c.height / (2 * c.fy))
        P = S.projection_matrix(0.001, 1000.0, fovx, fovy) @ cam.viewmat
        np.testing.assert_allclose(c.projmat, P, rtol=0, atol=0)
        np.testing.assert_array_equal(c.viewmat, cam.viewmat)
    odd = rescale_camera(S.Camera(1001, 667, 800.0, 800.0, 500.5, 333.5, cam.viewmat, cam.projmat), 4)
    assert (odd.width, odd.height) == (250, 166)  # truncated, as `(self.width * scaling_factor).to(torch.int64)`
    assert rescale_camera(cam, 1) is cam


def test_ground_truth_is_resized_then_composited_and_the_cached_planes_give_the_same_target():
    g = torch.Generator().manual_seed(0)
    rgba = torch.rand(64, 48, 4, generator=g)
    bg = torch.rand(3, generator=g)
    for d in (1, 2, 4):
        small = downscale_image(rgba, d)
        assert small.shape == (64 // d, 48 // d, 4)
        ref = composite_with_background(small, bg)  # vanilla_gs.py:870-881 on the RESIZED image
        a = small[..., 3:4]
        np.testing.assert_allclose(ref.numpy(), (a * small[..., :3] + (1 - a) * bg).numpy(), atol=1e-7)
        # TrainConfig.fused_target: alpha_ds * rgb_ds and 1 - alpha_ds cached, one addcmul per step
        fused = torch.addcmul((a * small[..., :3]).contiguous(), (1 - a).contiguous(), bg)
        assert (ref - fused).abs().max() <= 1.2e-7
    # resizing is bilinear without antialiasing (TF.resize(..., antialias=None) on tensors): factor 2 averages 2x2
    x = torch.arange(16.0).reshape(4, 4, 1)
    np.testing.assert_allclose(downscale_image(x, 2)[..., 0].numpy(), [[2.5, 4.5], [10.5, 12.5]])
    rgb = torch.rand(8, 8, 3, generator=g)
    assert composite_with_background(rgb, bg) is rgb  # no alpha channel: untouched
