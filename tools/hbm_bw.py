#!/usr/bin/env python3
"""Reference points for the HBM-bound phases: streaming write / copy rates of plain torch kernels on this box."""
import torch
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (135, 512, 2048):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")
    w = t(lambda: a.fill_(1.0)); c = t(lambda: b.copy_(a)); r = t(lambda: a.sum())
    print(f"{mb} MB: fill {mb/1024/w/1e3*1.024:.2f} TB/s   copy (r+w) {2*mb/1024/c/1e3*1.024:.2f} TB/s   read-reduce {mb/1024/r/1e3*1.024:.2f} TB/s")
