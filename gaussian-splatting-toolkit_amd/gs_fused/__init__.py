"""Fused caller-side ops next to the rasterizer (SURVEY.md 8f "next" rows): the
photometric loss head (f2).  Same native library and C ABI (`include/gsraster.h`)
as `rasterizer`; no CPU fallback."""
from .loss import L1SSIMLoss, l1_ssim_loss  # noqa: F401
