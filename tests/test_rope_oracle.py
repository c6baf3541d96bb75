"""The C restatement of the reference's RoPE-2D loop (oracle/rope_ref.c) against golden vectors produced by the
reference's own implementations (tests/golden/make_rope_goldens.py): its PyTorch fallback (pos_embed.py:112-159)
and its C++ CPU path (curope.cpp:11-47, compiled unmodified into oracle/_ref in the build container)."""
import pytest
import torch

from tests import util


@pytest.fixture(scope="module")
def goldens(golden_dir):
    return torch.load(golden_dir / "rope_goldens.pt")


def test_goldens_hold_both_reference_paths(goldens):
    assert goldens["has_cpp"], "goldens were generated without the reference C++ path"
    assert len(goldens["cases"]) >= 5


def test_oracle_matches_reference_cpp_path(goldens):
    for name, c in goldens["cases"].items():
        tok = c["tokens_BHND"].transpose(1, 2).contiguous()
        got = util.rope_oracle(tok, c["positions"], c["base"], c["F0"])
        err = float((got - c["out_cpp_BNHD"]).abs().max())
        assert err <= 1e-6, (name, err)             # same loop, same libm: bit-identical in practice
        assert c["roundtrip_cpp_maxerr"] < 2e-6


def test_oracle_matches_reference_pytorch_fallback(goldens):
    for name, c in goldens["cases"].items():
        tok = c["tokens_BHND"].transpose(1, 2).contiguous()
        got = util.rope_oracle(tok, c["positions"], c["base"], c["F0"]).transpose(1, 2)
        err = float((got - c["out_fallback_BHND"]).abs().max())
        assert err <= 1e-5, (name, err)             # the two reference paths agree to 4e-6


def test_oracle_inverse_and_strides():
    gen = torch.Generator().manual_seed(0)
    big = torch.randn(2, 10, 3, 4, 32, generator=gen)            # (B,N,3,H,D): q/k/v interleaved like blocks.py:97
    view = big[:, :, 1]                                          # strided [B,N,H,D] view, stride(2)=D
    pos = torch.randint(0, 9, (2, 10, 2), generator=gen)
    fwd = util.rope_oracle(view, pos, 100.0, 1.0)
    back = util.rope_oracle(fwd, pos, 100.0, -1.0)
    assert float((back - view).abs().max()) < 2e-6
    # position 0 is the identity; norms of (u,v) pairs are preserved
    zero = util.rope_oracle(view, torch.zeros_like(pos), 100.0, 1.0)
    assert torch.equal(zero, view.contiguous())
    assert float((fwd.pow(2).sum(-1) - view.pow(2).sum(-1)).abs().max()) < 1e-4


def test_oracle_matches_vggt_reference_rope(goldens):
    """VGGT's RotaryPositionEmbedding2D (vggt/layers/rope.py:62-188) is the same rotation, out of place."""
    for name, c in goldens["vggt"].items():
        tok = c["tokens_BHND"].transpose(1, 2).contiguous()
        got = util.rope_oracle(tok, c["positions"], c["frequency"], 1.0).transpose(1, 2)
        assert float((got - c["out_BHND"]).abs().max()) <= 1e-5, name


def test_torch_restatement_of_the_fallback_matches_the_reference_module(goldens):
    """oracle/rope_torch_ref.py (bench.py's second rope2d CPU baseline) restates the reference's fallback class
    pos_embed.py:112-159; the goldens hold that class's own outputs -- same torch ops, so bit-near identical."""
    from oracle import rope_torch_ref
    for name, c in goldens["cases"].items():
        if c["F0"] != 1.0:
            continue                                 # (the fallback module ignores F0 in its tables: pos_embed.py:120-129)
        cache = {}
        got = rope_torch_ref.rope2d_fallback(c["tokens_BHND"], c["positions"], c["base"], cache)
        assert float((got - c["out_fallback_BHND"]).abs().max()) <= 1e-6, name
        again = rope_torch_ref.rope2d_fallback(c["tokens_BHND"], c["positions"], c["base"], cache)   # cached tables
        assert torch.equal(again, got) and len(cache) == 1
