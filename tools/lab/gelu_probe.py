import sys, torch
sys.path.insert(0, '/root/repo')
import torch.nn.functional as F
from ufm_amd import hip
DEV='cuda'
bits = torch.arange(0, 1 << 16, dtype=torch.int32)
vals = (bits << 16).view(torch.float32)
vals = vals[torch.isfinite(vals) & (vals.abs() <= 8.0)]
n = (vals.numel() // 32) * 32
x = vals[:n].reshape(-1, 32).contiguous()
M = x.shape[0]
A = torch.stack([x.bfloat16(), torch.zeros_like(x).bfloat16()])
W = torch.stack([torch.eye(32).bfloat16(), torch.zeros(32, 32).bfloat16()])
out = torch.full((M, 32), 7.0, device=DEV)
hip.gemm_x3(A.to(DEV), W.to(DEV), M, 32, 32, out, torch.zeros(256, device=DEV), act=1)
ref = F.gelu(x.double())
err = (out.cpu().double() - ref).abs()
i = err.argmax()
print("max abs err", err.max().item(), "at x", x.flatten()[i].item())
pos = x >= 0
rel = (err / ref.abs().clamp_min(1e-300))
print("max rel err x>=2^-10", rel[(x >= 2.0**-10)].max().item())
print("max rel err x in [-1, -2^-10]", rel[(x <= -2.0**-10) & (x >= -1)].max().item())
print("max rel err x in [-3,-1]", rel[(x <= -1) & (x >= -3)].max().item())
print("max abs err x<-3", err[x < -3].max().item())
