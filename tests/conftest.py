import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gaussian-splatting-toolkit_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def tune():
    """Override rows of the dispatch rules' tuning table (rasterizer/cuda/_tuning.py) for one test:
    ``tune(two_round="0", depth_segments=5)`` REPLACES the overrides; everything is restored afterwards."""
    from rasterizer.cuda import _tuning

    saved = _tuning.overrides()

    def _set(**rows):
        _tuning.set_overrides(rows)

    yield _set
    _tuning.set_overrides(saved)
