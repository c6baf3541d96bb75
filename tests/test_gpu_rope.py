"""HIP RoPE-2D (through the C ABI) vs the reference-generated goldens and the C oracle."""
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
TOL = 1e-5   # BASELINE.md parity gate for curope (the reference's two own paths differ by 4e-6)


@pytest.fixture(scope="module")
def goldens(golden_dir):
    return torch.load(golden_dir / "rope_goldens.pt")


def test_goldens_fp32(hip_lib, goldens):
    import spfsplatv2_amd as spf
    for name, c in goldens["cases"].items():
        buf = c["tokens_BHND"].transpose(1, 2).contiguous().cuda()           # (B,N,H,D)-major buffer
        tok = buf.transpose(1, 2)                                             # the module sees [B,H,N,D]
        out = spf.cuRoPE2D(freq=c["base"], F0=c["F0"])(tok, c["positions"].cuda())
        assert out is tok                                                     # in place, same object (curope2d.py:40)
        assert float((out.cpu() - c["out_fallback_BHND"]).abs().max()) <= TOL, name
        assert float((buf.cpu() - c["out_cpp_BNHD"]).abs().max()) <= TOL, name


def test_strided_qkv_view_and_neighbours_untouched(hip_lib):
    import spfsplatv2_amd as spf
    gen = torch.Generator().manual_seed(1)
    B, N, H, D = 3, 258, 12, 64
    qkv = torch.randn(B, N, 3, H, D, generator=gen)
    pos = torch.randint(0, 18, (B, N, 2), generator=gen)
    want_q = util.rope_oracle(qkv[:, :, 0], pos, 100.0, 1.0)
    dev = qkv.cuda().transpose(1, 3)                # (B,H,3,N,D) view, exactly croco/blocks.py:97-98
    q = dev[:, :, 0]                                # [B,H,N,D] non-contiguous
    spf.cuRoPE2D()(q, pos.cuda())
    back = dev.transpose(1, 3).cpu()
    assert float((back[:, :, 0] - want_q).abs().max()) <= TOL
    assert torch.equal(back[:, :, 1], qkv[:, :, 1]) and torch.equal(back[:, :, 2], qkv[:, :, 2])


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 2e-2)])
def test_half_types(hip_lib, dtype, tol):
    import spfsplatv2_amd as spf
    gen = torch.Generator().manual_seed(2)
    buf = torch.randn(2, 64, 16, 64, generator=gen)
    pos = torch.randint(0, 18, (2, 64, 2), generator=gen)
    want = util.rope_oracle(buf.to(dtype).float(), pos, 100.0, 1.0)
    t = buf.to(dtype).cuda()
    spf.rope_2d(t, pos.cuda(), 100.0, 1.0)
    assert float((t.float().cpu() - want).abs().max()) <= tol


def test_autograd_backward_is_inverse_rotation(hip_lib):
    import spfsplatv2_amd as spf
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(2, 20, 4, 32, generator=gen).cuda().requires_grad_(True)     # (B,N,H,D)
    pos = torch.randint(0, 12, (2, 20, 2), generator=gen)
    w = torch.randn(2, 20, 4, 32, generator=gen)
    y = spf.cuRoPE2D_func.apply(x.clone(), pos.cuda(), 100.0, 1.0)
    (y * w.cuda()).sum().backward()
    want = util.rope_oracle(w, pos, 100.0, -1.0)         # d/dx sum(w * R x) = R^T w = rotation by -angle
    assert float((x.grad.cpu() - want).abs().max()) <= TOL


def test_error_behaviour(hip_lib):
    import spfsplatv2_amd as spf
    t = torch.zeros(2, 4, 3, 16, device="cuda")
    p = torch.zeros(2, 4, 2, dtype=torch.int64, device="cuda")
    with pytest.raises(RuntimeError, match="4 dimensions"):
        spf.rope_2d(t[0], p, 100.0, 1.0)
    with pytest.raises(RuntimeError, match="seq_length differs"):
        spf.rope_2d(t, p[:, :3], 100.0, 1.0)
    with pytest.raises(RuntimeError, match="tokens are not contiguous"):
        spf.rope_2d(torch.zeros(2, 4, 3, 32, device="cuda")[..., :16], p, 100.0, 1.0)       # stride(2) != D
    with pytest.raises(RuntimeError, match="positions are not contiguous"):
        spf.rope_2d(t, torch.zeros(2, 4, 4, dtype=torch.int64, device="cuda")[..., :2], 100.0, 1.0)
    with pytest.raises(RuntimeError, match="multiple of 4"):
        spf.rope_2d(torch.zeros(2, 4, 3, 6, device="cuda"), p, 100.0, 1.0)


def test_vggt_rotary_embedding_dropin(hip_lib, goldens):
    """RotaryPositionEmbedding2D (vggt/layers/rope.py:62-188): out of place, head-major, vs the reference's output."""
    import spfsplatv2_amd as spf
    for name, c in goldens["vggt"].items():
        tok = c["tokens_BHND"].cuda().requires_grad_(True)
        out = spf.RotaryPositionEmbedding2D(frequency=c["frequency"])(tok, c["positions"].cuda())
        assert out is not tok and out.shape == tok.shape
        assert float((out.detach().cpu() - c["out_BHND"]).abs().max()) <= TOL, name
        assert torch.equal(tok.detach().cpu(), c["tokens_BHND"])                 # input untouched
        w = torch.randn(tok.shape, generator=torch.Generator().manual_seed(4))
        (out * w.cuda()).sum().backward()
        B, H, N, D = tok.shape
        want = util.rope_oracle(w.transpose(1, 2).contiguous(), c["positions"], c["frequency"], -1.0).transpose(1, 2)
        assert float((tok.grad.cpu() - want).abs().max()) <= TOL, name


@pytest.mark.parametrize("seed", range(24))
def test_random_shapes_vs_c_oracle(hip_lib, seed):
    """Seeded sweep far from the model's shapes: ragged token counts, 1..16 heads, every head size the kernel accepts
    (multiples of 4 up to 128), q/k/v-strided and contiguous buffers, both rotation directions, two bases, positions
    up to the thousands (|angle| ~ 1e3 rad: the float32 sincos argument reduction is what this checks)."""
    import spfsplatv2_amd as spf
    g = torch.Generator().manual_seed(9000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    B, N, H, D = ri(1, 5), ri(1, 700), ri(1, 16), 4 * ri(1, 32)
    base = [100.0, 10000.0][ri(0, 1)]
    fwd = [1.0, -1.0][ri(0, 1)]
    pmax = [17, 40, 3000][ri(0, 2)]
    pos = torch.randint(0, pmax + 1, (B, N, 2), generator=g)
    strided = bool(ri(0, 1))
    if strided:
        qkv = torch.randn(B, N, 3, H, D, generator=g)
        which = ri(0, 2)
        want = util.rope_oracle(qkv[:, :, which], pos, base, fwd)
        dev = qkv.cuda()
        spf.rope_2d(dev[:, :, which], pos.cuda(), base, fwd)      # [B,N,H,D] view: stride(1) = 3*H*D
        got = dev.cpu()
        for other in range(3):
            if other != which:
                assert torch.equal(got[:, :, other], qkv[:, :, other])
        got = got[:, :, which]
    else:
        tok = torch.randn(B, N, H, D, generator=g)
        want = util.rope_oracle(tok, pos, base, fwd)
        dev = tok.cuda()
        spf.rope_2d(dev, pos.cuda(), base, fwd)
        got = dev.cpu()
    # angles up to 3000 rad: one float32 ulp of the angle is 2.4e-4 rad, and libm (the oracle) and the device both
    # round pos * inv_freq before reducing it -- the gate scales with that, 1e-5 at the model's positions (<= 40)
    tol = TOL if pmax <= 40 else 1.5e-7 * pmax * float(want.abs().max()) + TOL
    assert float((got - want).abs().max()) <= tol, (B, N, H, D, base, fwd, pmax, strided)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("H", [1, 3, 4, 12, 16])
def test_head_loop_and_pair_launch(hip_lib, dtype, tol, H):
    """Every head-count remainder of the four-heads-per-lane loop, 16 bytes per lane for every dtype, and q and k of one
    qkv buffer rotated in ONE launch (`rope_2d_pair`) -- against the C oracle (curope.cpp:11-47 restated) on the values
    the tensors actually hold."""
    import spfsplatv2_amd as spf
    gen = torch.Generator().manual_seed(5 + H)
    B, N, D = 2, 37, 64
    qkv = torch.randn(B, N, 3, H, D, generator=gen).to(dtype)
    pos = torch.randint(0, 18, (B, N, 2), generator=gen)
    want_q = util.rope_oracle(qkv[:, :, 0].float(), pos, 100.0, 1.0)
    want_k = util.rope_oracle(qkv[:, :, 1].float(), pos, 100.0, 1.0)
    dq = qkv.to("cuda")
    spf.rope_2d_pair(dq[:, :, 0], dq[:, :, 1], pos.to("cuda"), 100.0, 1.0)
    assert float((dq[:, :, 0].float().cpu() - want_q).abs().max()) < tol * max(1.0, float(want_q.abs().max()))
    assert float((dq[:, :, 1].float().cpu() - want_k).abs().max()) < tol * max(1.0, float(want_k.abs().max()))
    assert torch.equal(dq[:, :, 2].cpu(), qkv[:, :, 2])                     # v untouched
    single = qkv.to("cuda")
    spf.rope_2d(single[:, :, 0], pos.to("cuda"), 100.0, 1.0)
    assert torch.equal(single[:, :, 0], dq[:, :, 0])                        # pair launch == single launch, bit for bit
