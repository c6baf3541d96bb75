"""PyTorch restatement of the reference's RoPE-2D FALLBACK module (TEST INFRASTRUCTURE ONLY: the checker and the
`cpu_baseline` leg of bench.py's rope2d line; never imported by the product).

Follows /root/reference/src/model/encoder/backbone/croco/pos_embed.py:112-159 (class RoPE2D, the "slow pytorch version"
the reference falls back to when its compiled curope module is missing): out of place, tokens [B,H,N,D]; the feature
axis is split into a y half and an x half; for each half a cos / sin table [max_pos + 1, D/2] is built from
inv_freq = base^(-2i / (D/2)) (pos_embed.py:120-129: the D/4 frequencies repeated twice), gathered by the token's
position with an embedding lookup (pos_embed.py:136-139) and applied as  t * cos + rotate_half(t) * sin  with
rotate_half(t) = (-t2, t1) (pos_embed.py:131-134,140).  The table cache of the reference (keyed by D, length, device,
dtype) is a dict passed in by the caller here, so that a timing loop pays for the tables once, as the module does.

Pinned by tests/test_rope_oracle.py against tests/golden/rope_goldens.pt (outputs of the reference's own class).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _tables(half_dim: int, n_pos: int, base: float, device, dtype, cache: dict | None):
    key = (half_dim, n_pos, str(device), dtype)
    if cache is not None and key in cache:
        return cache[key]
    exponent = torch.arange(0, half_dim, 2, device=device).float() / half_dim
    inv_freq = 1.0 / (base ** exponent)                                   # [half_dim / 2]
    angle = torch.outer(torch.arange(n_pos, device=device, dtype=inv_freq.dtype), inv_freq).to(dtype)
    angle = torch.cat((angle, angle), dim=-1)                             # [n_pos, half_dim]
    out = (angle.cos(), angle.sin())
    if cache is not None:
        cache[key] = out
    return out


def _rope_1d(t: torch.Tensor, pos: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    c = F.embedding(pos, cos)[:, None]                                    # [B,1,N,half]
    s = F.embedding(pos, sin)[:, None]
    t1, t2 = t[..., : t.shape[-1] // 2], t[..., t.shape[-1] // 2:]
    return t * c + torch.cat((-t2, t1), dim=-1) * s


def rope2d_fallback(tokens_bhnd: torch.Tensor, positions: torch.Tensor, base: float = 100.0,
                    cache: dict | None = None) -> torch.Tensor:
    """tokens [B,H,N,D], positions [B,N,2] int64 (y, x) -> new tensor [B,H,N,D]."""
    assert tokens_bhnd.shape[-1] % 2 == 0 and positions.dim() == 3 and positions.shape[-1] == 2
    half = tokens_bhnd.shape[-1] // 2
    cos, sin = _tables(half, int(positions.max()) + 1, base, tokens_bhnd.device, tokens_bhnd.dtype, cache)
    ty, tx = tokens_bhnd.chunk(2, dim=-1)
    return torch.cat((_rope_1d(ty, positions[..., 0], cos, sin), _rope_1d(tx, positions[..., 1], cos, sin)), dim=-1)
