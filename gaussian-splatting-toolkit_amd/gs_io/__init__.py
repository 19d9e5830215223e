"""On-disk formats either side of the rasterizer."""
from .ply import read_gaussian_ply, write_gaussian_ply  # noqa: F401
