#!/usr/bin/env python3
"""Error pattern of ufm_attention_bf16 (scale == 0 form) vs fp64 for a few shapes: which query rows / d columns are off."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ufm_amd import hip  # noqa: E402


def attn_ref(qkv, B, N, H, scale):
    q, k, v = qkv.double().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    p = torch.softmax((q @ k.transpose(-1, -2)) * scale, dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B * N, H * 64)


for variant in (0, 1):
    for B, N, H in ((1, 300, 1), (10, 300, 16), (20, 200, 16), (8, 1370, 16), (8, 600, 12)):
        g = torch.Generator().manual_seed(N)
        qkv = (torch.randn(B * N, 3 * H * 64, generator=g) * 1.5).bfloat16().float()
        c = 0.125 * 1.4426950408889634
        pre = qkv.clone()
        pre[:, : H * 64] = (pre[:, : H * 64] * c).bfloat16().float()
        ref = attn_ref(torch.cat([pre[:, : H * 64] / c, pre[:, H * 64 :]], 1), B, N, H, 0.125)
        out = torch.zeros(B * N, H * 64, device="cuda", dtype=torch.bfloat16)
        hip.lib().ufm_debug_set_attn_variant(variant)
        hip.attention(pre.cuda().bfloat16(), out, B, N, H, 0.0)
        err = (out.float().cpu().double() - ref).abs()
        bad_rows = (err.max(1).values > 3e-2).nonzero().flatten().tolist()
        bad_cols = (err.max(0).values > 3e-2).nonzero().flatten().tolist()
        out2 = torch.zeros_like(out)
        hip.attention(pre.cuda().bfloat16(), out2, B, N, H, 0.0)
        same = bool(torch.equal(out.view(torch.int16), out2.view(torch.int16)))
        print(f"variant {variant} B{B} N{N} H{H}: repeat-bitwise {same} max err {err.max():.4f}  bad rows {len(bad_rows)} {bad_rows[:12]}  bad cols {len(bad_cols)} {bad_cols[:12]}", flush=True)
