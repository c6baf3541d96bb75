"""Photometric MSE loss: oracle (oracle/loss_ref.py) pinned by vectors captured from the reference's own LossMse
(tests/golden/make_loss_goldens.py; loss_mse.py:36-51), and the fused HIP op against both.
Tolerances: loss 2e-6 relative (float32 sum of up to 1.6e5 squares), gradient 1e-7 absolute (values ~1e-5)."""
from pathlib import Path

import pytest
import torch

from oracle import loss_ref

GOLD = torch.load(Path(__file__).parent / "golden" / "loss_goldens.pt")


def _rel(a, b):
    return abs(float(a) - float(b)) / max(abs(float(b)), 1e-12)


@pytest.mark.parametrize("name", list(GOLD))
def test_oracle_matches_reference_loss(name):
    g = GOLD[name]
    loss, grad = loss_ref.mse_loss(g["prediction"], g["image"], g["weight"], g["global_step"], g["apply_after_step"])
    if float(g["loss"]) == 0.0:
        assert float(loss) == 0.0 and float(grad.abs().max()) == 0.0
    else:
        assert _rel(loss, g["loss"]) < 2e-6
        assert float((grad.float() - g["grad"]).abs().max()) < 1e-9 + 1e-6 * float(g["grad"].abs().max())


def test_loss_module_surface_and_cpu_refusal():
    from spfsplatv2_amd import loss as L
    m = L.LossMse(L.LossMseCfgWrapper(L.LossMseCfg(weight=0.5, apply_after_step=3)))
    assert m.name == "mse" and m.cfg.weight == 0.5 and len(list(m.parameters())) == 0
    x = torch.rand(1, 1, 3, 4, 4)
    assert float(m(x, x, None, 0)) == 0.0            # before apply_after_step: a constant 0 (loss_mse.py:44-46)
    with pytest.raises(RuntimeError, match="no CPU"):
        m(x, x, None, 5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(GOLD))
def test_hip_loss_matches_reference_vectors(hip_lib, name):
    from spfsplatv2_amd import loss as L
    g = GOLD[name]
    pred = g["prediction"].cuda().requires_grad_(True)
    img = g["image"].cuda().requires_grad_(True)
    m = L.LossMse(L.LossMseCfgWrapper(L.LossMseCfg(weight=g["weight"], apply_after_step=g["apply_after_step"])))
    loss = m(pred, img, None, g["global_step"])
    if float(g["loss"]) == 0.0:
        assert float(loss) == 0.0 and not loss.requires_grad
        return
    assert loss.shape == () and loss.dtype == torch.float32
    assert _rel(loss, g["loss"]) < 2e-6
    (3.0 * loss).backward()                              # upstream scalar is read on the device
    assert float((pred.grad.cpu() - 3.0 * g["grad"]).abs().max()) < 1e-7
    assert torch.equal(img.grad, -pred.grad)


@pytest.mark.gpu
def test_hip_loss_is_deterministic_and_takes_views(hip_lib):
    from spfsplatv2_amd import loss as L
    gen = torch.Generator().manual_seed(5)
    a = torch.rand(8, 4, 3, 256, 256, generator=gen).cuda()      # the bench's batch: 6.3 M floats, 1024 partial blocks
    b = torch.rand(8, 4, 3, 256, 256, generator=gen).cuda()
    l1, l2 = L.mse_loss(a, b), L.mse_loss(a, b)
    assert torch.equal(l1, l2)
    ref, _ = loss_ref.mse_loss(a, b, 1.0)
    assert _rel(l1, ref) < 2e-6
    # a non-contiguous prediction (channel-last view) gives the same loss and a gradient of the view's shape
    av = a.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3).requires_grad_(True)
    lv = L.mse_loss(av, b, 0.25)
    assert _rel(lv, 0.25 * ref) < 2e-6
    lv.backward()
    assert av.grad.shape == a.shape
    want = 0.25 * 2.0 / a.numel() * (a - b)
    assert float((av.grad - want).abs().max()) < 1e-12 + 1e-6 * float(want.abs().max())


@pytest.mark.gpu
def test_hip_loss_accepts_what_the_reference_expression_accepts(hip_lib):
    """bf16 inputs (autocast), broadcastable shapes and contiguous-but-unaligned slices are cast / expanded /
    copied on the host side instead of being refused; two calls never share scratch."""
    from spfsplatv2_amd import loss as L
    gen = torch.Generator().manual_seed(9)
    a = torch.rand(2, 3, 3, 17, 19, generator=gen).cuda()
    b = torch.rand(2, 3, 3, 17, 19, generator=gen).cuda()
    want = ((a - b) ** 2).mean()
    # odd-sized image, leading-dim slice starting at an address that is not a multiple of 16 bytes
    big_a, big_b = torch.rand(3, 3, 3, 17, 19, generator=gen).cuda(), torch.rand(3, 3, 3, 17, 19, generator=gen).cuda()
    sa, sb = big_a[1:], big_b[1:]
    assert sa.is_contiguous() and sa.data_ptr() % 16 != 0
    assert _rel(L.mse_loss(sa, sb), ((sa - sb) ** 2).mean()) < 2e-6
    # bf16 prediction: evaluated in float32, gradient comes back in bf16
    ah = a.bfloat16().requires_grad_(True)
    lh = L.mse_loss(ah, b)
    assert lh.dtype == torch.float32 and _rel(lh, ((ah.float() - b) ** 2).mean()) < 2e-6
    lh.backward()
    assert ah.grad.dtype == torch.bfloat16 and ah.grad.shape == a.shape
    # broadcast: one target image for every view
    tgt = b[:, :1].clone().requires_grad_(True)
    lb = L.mse_loss(a, tgt)
    assert _rel(lb, ((a - tgt) ** 2).mean()) < 2e-6
    lb.backward()
    wantg = (-2.0 / a.numel() * (a - tgt.detach())).sum(dim=1, keepdim=True)
    assert tgt.grad.shape == tgt.shape and float((tgt.grad - wantg).abs().max()) < 1e-9 + 1e-5 * float(wantg.abs().max())
    # concurrent streams: each call owns its partial sums
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        l1 = [L.mse_loss(a, b) for _ in range(20)]
    with torch.cuda.stream(s2):
        l2 = [L.mse_loss(big_a, big_b) for _ in range(20)]
    torch.cuda.synchronize()
    assert all(torch.equal(x, l1[0]) for x in l1) and all(torch.equal(x, l2[0]) for x in l2)
    assert _rel(l1[0], want) < 2e-6


@pytest.mark.gpu
def test_hip_loss_unit_gradient_from_the_forward_and_second_backward(hip_lib):
    """The forward writes the gradient for dL/dloss = 1; the first backward scales that buffer in place (a no-op launch
    for 1), a second backward through the same node (retain_graph) recomputes from the inputs."""
    from spfsplatv2_amd import loss as L
    gen = torch.Generator().manual_seed(11)
    a = torch.rand(2, 2, 3, 33, 31, generator=gen).cuda().requires_grad_(True)       # 12,276 floats: n % 4 == 0; tail below
    b = torch.rand(2, 2, 3, 33, 31, generator=gen).cuda()
    want = 0.5 * 2.0 / a.numel() * (a.detach() - b)
    l = L.mse_loss(a, b, 0.5)
    l.backward(retain_graph=True)
    g1 = a.grad.clone()
    assert float((g1 - want).abs().max()) <= 1e-6 * float(want.abs().max())
    a.grad = None
    (2.0 * l).backward()                                  # second time through the node, upstream 2
    assert float((a.grad - 2.0 * g1).abs().max()) <= 2e-6 * float(want.abs().max())
    # upstream != 1 on the FIRST backward: the in-place scaling pass; odd element count: the tail
    a2 = torch.rand(7, 5, 3, generator=gen).cuda().requires_grad_(True)               # 105 floats
    b2 = torch.rand(7, 5, 3, generator=gen).cuda()
    (L.mse_loss(a2, b2) * 0.125).backward()
    want2 = 0.125 * 2.0 / 105 * (a2.detach() - b2)
    assert float((a2.grad - want2).abs().max()) <= 1e-6 * float(want2.abs().max())
    # no backward coming: nothing extra is written, the value is the same
    with torch.no_grad():
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated()
        l_eval = L.mse_loss(a, b, 0.5)
        # (only the loss scalar and the block partials are live: no batch-sized unit-gradient buffer, although `a`
        #  requires grad -- needs_input_grad stays True under no_grad, the call site's grad mode decides)
        assert torch.cuda.memory_allocated() - before < a.numel() * 4
        assert torch.equal(l_eval, l.detach())
    # the process-wide dL/dloss = 1 (`unit_grad`): recognised by its storage, the backward launches nothing and hands
    # the forward's unit gradient on as it is
    a3 = a.detach().clone().requires_grad_(True)
    L.mse_loss(a3, b, 0.5).backward(gradient=L.unit_grad(a3.device))
    assert torch.equal(a3.grad, g1)
