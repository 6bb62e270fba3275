#!/usr/bin/env python3
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ufm_amd import hip
lib = hip.lib()
M, N, K, variant = (int(v) for v in sys.argv[1:5])
lib.ufm_debug_set_gemm_variant(variant)
A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * K**-0.5).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    hip.gemm_bf16(A, W, M, N, K, out)
torch.cuda.synchronize()
