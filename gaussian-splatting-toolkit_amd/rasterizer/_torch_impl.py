"""Small pure-PyTorch helpers the models import from ``rasterizer._torch_impl``.

The reference's module of this name is its forward-only PyTorch restatement of
the kernels (470 lines).  The models only import ``quat_to_rotmat`` from it
(gs_toolkit/models/vanilla_gs.py:13, used when splitting Gaussians), so that
is what is provided here.  Nothing in this file is a fallback for the HIP
kernels: there is deliberately no torch implementation of projection, binning
or compositing in the product package (the CPU checker lives in ``oracle/``).
"""
import torch
import torch.nn.functional as F
from torch import Tensor


def normalized_quat_to_rotmat(quat: Tensor) -> Tensor:
    """Rotation matrices [...,3,3] from unit quaternions [...,4] in (w,x,y,z)."""
    assert quat.shape[-1] == 4, quat.shape
    w, x, y, z = torch.unbind(quat, dim=-1)
    rows = (
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y),
    )
    return torch.stack(rows, dim=-1).reshape(quat.shape[:-1] + (3, 3))


def quat_to_rotmat(quat: Tensor) -> Tensor:
    """As above, normalising the quaternion first."""
    assert quat.shape[-1] == 4, quat.shape
    return normalized_quat_to_rotmat(F.normalize(quat, dim=-1))
