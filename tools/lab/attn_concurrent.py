#!/usr/bin/env python3
"""Bitwise repeatability of each trunk kernel (UFM-Base shapes) while another stream runs another trunk kernel."""
import os, sys, threading, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufm_amd import hip
hip.lib()
torch.manual_seed(0)
M, D = 10960, 768
dev = "cuda"
def mk_attn(b, n, h):
    qkv = torch.randn(b * n, 3 * h * 64, device=dev).bfloat16()
    return lambda out: hip.attention(qkv, out, b, n, h, 0.0), (b * n, h * 64), torch.bfloat16
def mk_gemm(m, n, k, res, act):
    A = torch.randn(m, k, device=dev).bfloat16(); W = (torch.randn(n, k, device=dev) * 0.03).bfloat16()
    bias = torch.randn(n, device=dev); gamma = torch.randn(n, device=dev)
    R = torch.randn(m, n, device=dev) if res else None
    if res:
        return lambda out: hip.gemm_bf16(A, W, m, n, k, out, bias=bias, gamma=gamma, res=R), (m, n), torch.float32
    return lambda out: hip.gemm_bf16(A, W, m, n, k, out, bias=bias, act=act), (m, n), torch.bfloat16
def mk_ln(m, d):
    x = torch.randn(m, d, device=dev); w = torch.randn(d, device=dev); b = torch.randn(d, device=dev)
    return lambda out: hip.layernorm(x, d, None, m, d, w, b, 1e-6, out), (m, d), torch.bfloat16
ops = {
    "attn_enc": mk_attn(8, 1370, 12), "attn_info": mk_attn(4, 2740, 12),
    "gemm_qkv": mk_gemm(M, 3 * D, D, False, 0), "gemm_proj": mk_gemm(M, D, D, True, 0),
    "gemm_fc1": mk_gemm(M, 4 * D, D, False, 1), "gemm_fc2": mk_gemm(M, D, 4 * D, True, 0), "ln": mk_ln(M, D),
}
side = torch.cuda.Stream()
TARGETS = os.environ.get("TARGETS", "").split(",") if os.environ.get("TARGETS") else list(ops)
LOADS = os.environ.get("LOADS", "").split(",") if os.environ.get("LOADS") else list(ops)
REPS = int(os.environ.get("REPS", "12"))
for tname, (tfn, tshape, tdt) in ops.items():
    if tname not in TARGETS: continue
    ref = torch.zeros(tshape, device=dev, dtype=tdt); tfn(ref); torch.cuda.synchronize()
    for lname, (lfn, lshape, ldt) in ops.items():
        if lname not in LOADS: continue
        lout = torch.zeros(lshape, device=dev, dtype=ldt)
        stop = False
        def load():
            with torch.cuda.stream(side):
                while not stop:
                    for _ in range(10): lfn(lout)
                    side.synchronize()
        t = threading.Thread(target=load); t.start()
        bad = 0; info = ""
        for rep in range(REPS):
            out = torch.zeros(tshape, device=dev, dtype=tdt)
            tfn(out); torch.cuda.synchronize()
            if not torch.equal(out.view(torch.uint8), ref.view(torch.uint8)):
                bad += 1
                d = (out.float() - ref.float()).abs()
                rows = (d.max(1).values > 0).nonzero().flatten()
                info = f"max {d.max().item():.4g} rows {rows.numel()} [{rows[0].item()}..{rows[-1].item()}]"
        stop = True; t.join()
        print(f"{tname:10s} under {lname:10s}: mismatching {bad}/{REPS} {info}", flush=True)
