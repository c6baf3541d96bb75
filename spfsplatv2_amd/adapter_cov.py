"""World-space covariance of the adapter's Gaussians, for callers that still want ``Gaussians.covariances``
(reference: src/model/encoder/common/gaussians.py:8-44; quaternion in scipy xyzw order).  Plain torch: nothing on the
decoder path reads it."""
import torch
from torch import Tensor


def quaternion_to_matrix(q: Tensor, eps: float = 1e-8) -> Tensor:
    i, j, k, r = torch.unbind(q, dim=-1)
    two_s = 2 / ((q * q).sum(dim=-1) + eps)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(*q.shape[:-1], 3, 3)


def build_covariance(scale: Tensor, rotation_xyzw: Tensor) -> Tensor:
    R = quaternion_to_matrix(rotation_xyzw)
    RS = R * scale[..., None, :]
    return RS @ RS.transpose(-1, -2)
