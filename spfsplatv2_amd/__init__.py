"""spfsplatv2_amd -- MI355X (gfx950) native Gaussian-splat rasterizer + RoPE-2D behind SPFSplatV2's
decoder / curope call surfaces.  All compute lives in libspfsplat_hip.so (C ABI: include/spfsplat_hip.h).
"""
from . import _lib, hostbind
from .decoder import (DECODERS, camera_tensors, Decoder, DecoderOutput, DecoderSplattingCUDA, DecoderSplattingCUDACfg,
                      DecoderSplattingHIP, Gaussians, get_decoder, get_fov, get_projection_matrix, render_cuda,
                      orthographic_camera, render_cuda_orthographic, render_views)
from .rasterizer import (CallRecord, GaussianRasterizationSettings, GaussianRasterizer, PairBudget, camera_forward,
                         last_forward_stats, last_plan_flags, plan_flags, plan_pair_budget, rasterize_batch,
                         render_batch, sh_band4_default)
from .loss import Loss, LossMse, LossMseCfg, LossMseCfgWrapper, mse_loss, unit_grad
from .rope import (PositionGetter, RoPE2D, RotaryPositionEmbedding2D, append_token_position, cuRoPE2D, cuRoPE2D_func,
                   rope_2d, rope_2d_head_major, rope_2d_pair)

__all__ = [
    "DECODERS", "Decoder", "DecoderOutput", "DecoderSplattingCUDA", "DecoderSplattingCUDACfg",
    "DecoderSplattingHIP", "Gaussians", "get_decoder", "get_fov", "get_projection_matrix", "render_cuda",
    "render_cuda_orthographic", "render_views", "GaussianRasterizationSettings", "GaussianRasterizer",
    "last_forward_stats", "PairBudget", "plan_pair_budget", "last_plan_flags", "plan_flags", "CallRecord", "sh_band4_default", "orthographic_camera", "rasterize_batch", "render_batch", "camera_forward", "camera_tensors", "Loss", "LossMse", "LossMseCfg", "LossMseCfgWrapper", "mse_loss", "unit_grad", "PositionGetter", "append_token_position", "RoPE2D", "RotaryPositionEmbedding2D", "cuRoPE2D", "cuRoPE2D_func", "rope_2d", "rope_2d_head_major", "rope_2d_pair", "hostbind",
]
