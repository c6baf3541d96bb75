"""RoPE-2D kernel throughput vs the HBM roofline (run on the GPU box).  Shapes from BASELINE.md:
(B,N,H,D) = (48,256,16,64) encoder self-attention and (32,258,12,64) decoder, as strided q views of a qkv buffer."""
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import spfsplatv2_amd as spf
from spfsplatv2_amd import _lib

out = []
for (B, N, H, D) in ((48, 256, 16, 64), (32, 258, 12, 64), (256, 258, 12, 64)):
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        qkv = torch.randn(B, N, 3, H, D, device="cuda", dtype=dt)
        q = qkv[:, :, 0]                                  # [B,N,H,D] strided view, stride(2) = D
        pos = torch.randint(0, 18, (B, N, 2), device="cuda")
        for _ in range(5):
            spf.rope_2d(q, pos, 100.0, 1.0)
        _lib.stage_timing_enable(["rope2d"])
        reps = 50
        for _ in range(reps):
            spf.rope_2d(q, pos, 100.0, 1.0)
        torch.cuda.synchronize()
        ms, cnt = _lib.stage_times()["rope2d"]
        _lib.stage_timing_enable(False)
        byts = 2 * B * N * H * D * q.element_size() + 16 * B * N
        out.append({"shape": [B, N, H, D], "dtype": str(dt).split(".")[-1], "us": round(1e3 * ms / cnt, 2),
                    "GBs": round(byts / (ms / cnt * 1e-3) / 1e9, 1), "frac_of_8TBs": round(byts / (ms / cnt * 1e-3) / 8e12, 4)})
        print(out[-1], flush=True)
        # q and k of the same qkv buffer in one launch (croco/blocks.py:102-104 rotates them with two calls)
        k = qkv[:, :, 1]
        for _ in range(5):
            spf.rope_2d_pair(q, k, pos, 100.0, 1.0)
        _lib.stage_timing_enable(["rope2d"])
        for _ in range(reps):
            spf.rope_2d_pair(q, k, pos, 100.0, 1.0)
        torch.cuda.synchronize()
        ms, cnt = _lib.stage_times()["rope2d"]
        _lib.stage_timing_enable(False)
        byts = 4 * B * N * H * D * q.element_size() + 16 * B * N
        out.append({"shape": [B, N, H, D], "dtype": str(dt).split(".")[-1], "tensors": "q+k, one launch",
                    "us": round(1e3 * ms / cnt, 2), "GBs": round(byts / (ms / cnt * 1e-3) / 1e9, 1),
                    "frac_of_8TBs": round(byts / (ms / cnt * 1e-3) / 8e12, 4)})
        print(out[-1], flush=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "rope_bench.json"), "w"), indent=1)
