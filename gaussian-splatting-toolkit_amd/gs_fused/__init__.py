"""Fused caller-side ops next to the rasterizer (SURVEY.md 8f "next" rows): the
photometric loss head (f2), the one-launch Adam step (f1) and SH colours from
split coefficients and the per-Gaussian activations (f4).  Same native library and C ABI (`include/gsraster.h`)
as `rasterizer`; no CPU fallback."""
from .loss import L1SSIMLoss, depth_l1_loss, l1_loss, l1_ssim_loss  # noqa: F401
from .adam import FusedAdam  # noqa: F401
from .sh import sh_backward_views, spherical_harmonics_split  # noqa: F401
from .activations import activate_gaussians, densify_stats_  # noqa: F401
from .rgbd import rasterize_gaussians_rgbd  # noqa: F401
from .refine import RefineConfig, adam_moments, refine_gaussians, refinement_branch, swap_parameters  # noqa: F401
from .render import DensifyStats, ListCapacity, ViewSpec, render_gaussians  # noqa: F401
