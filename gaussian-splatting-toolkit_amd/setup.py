"""Installs the drop-in `rasterizer` package (plus `gs_fused`, `gs_io`, `harness`)
with its native library.  `pip install -e .` from this directory, after or instead of
uninstalling the reference's CUDA `rasterizer` wheel:

    pip install --no-build-isolation -e gaussian-splatting-toolkit_amd

The HIP sources are compiled by `make -C csrc` (hipcc, --offload-arch=gfx950; no GPU
needed to build); the resulting `rasterizer/cuda/libgsraster.so` ships as package data.
"""
import os
import subprocess

from setuptools import find_packages, setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))


class BuildWithNativeLibrary(build_py):
    def run(self):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "csrc"), "-j8"])
        super().run()


setup(
    name="rasterizer",
    version="0.1.2",  # the reference's rasterizer/version.py
    description="MI355X-native (gfx950) differentiable Gaussian-splatting rasterizer; drop-in for the "
                "Gaussian-Splatting-Toolkit `rasterizer` package",
    packages=find_packages(where=HERE, include=["rasterizer", "rasterizer.*", "gs_fused", "gs_io", "harness"]),
    package_dir={"": "."},
    package_data={"rasterizer.cuda": ["libgsraster.so"]},
    python_requires=">=3.8",
    install_requires=["torch", "numpy"],
    cmdclass={"build_py": BuildWithNativeLibrary},
)
