"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's photometric MSE loss.

Follows /root/reference/src/loss/loss_mse.py:36-51: ``weight * ((prediction - image) ** 2).mean()`` and the 0 before
``apply_after_step``; the gradient is autograd's.  Pinned by tests/golden/loss_goldens.pt, captured by importing the
reference's own ``LossMse`` (tests/golden/make_loss_goldens.py).  Only tests/ and __graft_entry__.smoke() may import
this module; the product never does.
"""
import torch


def mse_loss(prediction: torch.Tensor, image: torch.Tensor, weight: float, global_step: int = 0,
             apply_after_step: int = 0, dtype=torch.float64):
    """Returns (loss, dloss/dprediction) computed in `dtype` on the CPU."""
    p = prediction.detach().to("cpu", dtype).requires_grad_(True)
    t = image.detach().to("cpu", dtype)
    if global_step < apply_after_step:       # loss_mse.py:44-46
        return torch.zeros((), dtype=dtype), torch.zeros_like(p)
    delta = p - t                            # loss_mse.py:48
    loss = weight * (delta ** 2).mean()      # loss_mse.py:51
    loss.backward()
    return loss.detach(), p.grad
