#!/usr/bin/env python3
"""Round 6 pricing: what do the two quantisation losses of ufm_attention_bf16 cost, and what would the proposed fixes buy?
(a) the encoder's ragged last query unit: N = 1370 = 5 units of 256 rows + 90 rows computed as a sixth full unit (1536 rows, 10.8 % masked).
    Measured: the full launch; the same launch restricted to the 5 whole units (Nq = 1280 of 1370 keys: the strided entry point); the 90 tail rows
    alone on the four-wave kernel and on the two-wave kernel (128-row units, ufm_debug_set_attn_variant bit 0) -- on a second stream beside the
    whole-unit launch (what a two-launch form would run) and by themselves.
(b) the joint attention: 1056 units (B = 8) / 528 (one micro-batch of 4) on 256 persistent workgroups: the launch against its own whole-round part."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufm_amd import hip
lib = hip.lib()
DEV = "cuda"


def t_us(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


torch.manual_seed(0)
print("## (a) encoder attention, N = 1370, H = 16 (q pre-scaled: the persistent LDS-DMA kernel)")
for B in (16, 8):
    N, H = 1370, 16
    ld = 3 * H * 64
    qkv = (torch.randn(B * N, ld, device=DEV) * 0.5).bfloat16()
    out = torch.empty(B * N, H * 64, device=DEV, dtype=torch.bfloat16)
    q, k, v = qkv, qkv[:, H * 64:], qkv[:, 2 * H * 64:]
    full = t_us(lambda: hip.attention(qkv, out, B, N, H, 0.0))
    whole = t_us(lambda: hip.attention_strided(q, ld, N, k, v, ld, N, out, H * 64, N, B, 1280, N, H))
    qt, ot = qkv[1280:], out[1280:]
    tail4 = t_us(lambda: hip.attention_strided(qt, ld, N, k, v, ld, N, ot, H * 64, N, B, 90, N, H))
    lib.ufm_debug_set_attn_variant(1)
    tail2 = t_us(lambda: hip.attention_strided(qt, ld, N, k, v, ld, N, ot, H * 64, N, B, 90, N, H))
    lib.ufm_debug_set_attn_variant(0)
    side = torch.cuda.Stream()

    def two_launch():
        side.wait_stream(torch.cuda.current_stream())
        hip.attention_strided(q, ld, N, k, v, ld, N, out, H * 64, N, B, 1280, N, H)
        with torch.cuda.stream(side):
            lib.ufm_debug_set_attn_variant(1)
            hip.attention_strided(qt, ld, N, k, v, ld, N, ot, H * 64, N, B, 90, N, H)
            lib.ufm_debug_set_attn_variant(0)
        torch.cuda.current_stream().wait_stream(side)
    both = t_us(two_launch)
    units = B * H * 6
    print(f"B = {B:2d}: {units} units of 256 rows = {units / 256:.2f} rounds.  full launch {full:7.1f} us | 5 whole units per (image, head) only {whole:7.1f} us ({100 * (whole / full - 1):+5.1f} %: "
          f"the upper bound of any fix) | the 90-row tails alone: four-wave kernel {tail4:6.1f} us, two-wave kernel {tail2:6.1f} us | whole units + two-wave tails on a second stream {both:7.1f} us ({100 * (both / full - 1):+5.1f} %)", flush=True)

print("## (b) joint attention, N = 2738, H = 12: units = pairs x 12 x 11 on 256 persistent workgroups")
for B in (8, 4, 5):
    N, H = 2738, 12
    ld = 3 * H * 64
    qkv = (torch.randn(B * N, ld, device=DEV) * 0.5).bfloat16()
    out = torch.empty(B * N, H * 64, device=DEV, dtype=torch.bfloat16)
    full = t_us(lambda: hip.attention(qkv, out, B, N, H, 0.0))
    units = B * H * 11
    rounds = units / 256
    print(f"B = {B}: {units} units = {rounds:.3f} rounds -> {-(-units // 256)} unit times; launch {full:7.1f} us = {full / -(-units // 256):6.1f} us per round; at {rounds:.3f} rounds of that: {full / -(-units // 256) * rounds:7.1f} us "
          f"({100 * (rounds / -(-units // 256) - 1):+5.1f} % = what perfect packing of a LONE launch would save; in the two-stream pipeline the other stream's workgroups take the idle CUs)", flush=True)
