#!/usr/bin/env python3
"""What clock and board power does the chip hold under each hot kernel?  Back-to-back launches of one kernel for ~3 s while a
thread polls `rocm-smi` (sclk, average power, power cap).  The MFMA-dense loops are power-limited (MI355X_MICROARCH.md, DVFS
give-back): the clock the chip holds, not the instruction stream, sets their rate."""
import os, subprocess, sys, threading, time, re, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufm_amd import hip
lib = hip.lib()
DEV = "cuda"


def poll(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showmaxpower"], capture_output=True, text=True, timeout=10).stdout
            sclk = re.search(r"sclk clock level:?\s*\d*:?\s*\((\d+)Mhz\)", r)
            pw = re.search(r"(?:Average|Current Socket) Graphics Package Power \(W\):\s*([\d.]+)", r)
            cap = re.search(r"Max Graphics Package Power \(W\):\s*([\d.]+)", r)
            out.append((int(sclk.group(1)) if sclk else None, float(pw.group(1)) if pw else None, float(cap.group(1)) if cap else None))
        except Exception as exc:  # noqa: BLE001
            out.append((None, None, repr(exc)[:60]))
        time.sleep(0.2)


def under_load(name, fn, seconds=3.0, per_sync=20):
    stop, samples = threading.Event(), []
    th = threading.Thread(target=poll, args=(stop, samples))
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    th.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(per_sync):
            fn()
        torch.cuda.synchronize()
        n += per_sync
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    s = [x for x in samples[2:] if x[0]]
    clk = sorted(x[0] for x in s) if s else [0]
    pw = sorted(x[1] for x in s if x[1]) or [0]
    cap = next((x[2] for x in s if isinstance(x[2], float)), None)
    print(f"{name:44s} {dt / n * 1e6:8.1f} us/launch   sclk median {clk[len(clk)//2]:5d} MHz (min {clk[0]}, max {clk[-1]})   power median {pw[len(pw)//2]:6.1f} W (cap {cap})   [{len(s)} samples]", flush=True)
    if not s:
        print("   raw:", samples[:3])


M = 21920
A = torch.randn(M, 4096, device=DEV).bfloat16()
W1 = (torch.randn(4096, 1024, device=DEV) * 0.03).bfloat16()
W2 = (torch.randn(1024, 4096, device=DEV) * 0.03).bfloat16()
bias4, bias1 = torch.randn(4096, device=DEV), torch.randn(1024, device=DEV)
o_bf = torch.empty(M, 4096, device=DEV, dtype=torch.bfloat16)
x32 = torch.randn(M, 1024, device=DEV)
under_load("idle (no launches)", lambda: None, 1.5)
under_load("gemm fc1 M21920 N4096 K1024 bf16 GELU", lambda: hip.gemm_bf16(A[:, :1024], W1, M, 4096, 1024, o_bf, bias=bias4, act=hip.ACT_GELU, lda=4096))
under_load("gemm fc2 M21920 N1024 K4096 f32 +=", lambda: hip.gemm_bf16(A, W2, M, 1024, 4096, x32, bias=bias1, res=x32))
As = A[:2048].contiguous()
under_load("gemm fc2 on 32 CUs (M2048: 32 tiles)", lambda: (lib.ufm_debug_set_gemm_variant(4), lib.ufm_debug_set_gemm_tile_rows(256), hip.gemm_bf16(As, W2, 2048, 1024, 4096, x32[:2048], bias=bias1, res=x32[:2048]), lib.ufm_debug_set_gemm_variant(0), lib.ufm_debug_set_gemm_tile_rows(0)))
qkv = torch.randn(16 * 1370, 3 * 1024, device=DEV).bfloat16()
ao = torch.empty(16 * 1370, 1024, device=DEV, dtype=torch.bfloat16)
under_load("attention B16 N1370 H16 (pre-scaled q)", lambda: hip.attention(qkv, ao, 16, 1370, 16, 0.0))
zero = torch.zeros(512, device=DEV)
xc = torch.randn(2, 8, 148, 148, 256, device=DEV).bfloat16()
wc = (torch.randn(2, 256, 3, 3, 256, device=DEV) * 0.02).bfloat16()
oc = torch.empty(2, 8, 148, 148, 256, device=DEV, dtype=torch.bfloat16)
under_load("conv bf16x3 148^2 256->256 3x3 B8 (8-phase)", lambda: hip.conv2d_x3(xc, 8, 148, 148, 256, wc, 256, 3, 3, 1, 1, oc, zero))
xl = torch.randn(M, 1024, device=DEV)
g1, b1 = torch.ones(1024, device=DEV), torch.zeros(1024, device=DEV)
ol = torch.empty(M, 1024, device=DEV, dtype=torch.bfloat16)
under_load("layernorm 21920 x 1024 -> bf16", lambda: hip.layernorm(xl, 1024, None, M, 1024, g1, b1, 1e-6, ol))

# numerics "precise": the split-format trunk kernels
def _split(x):
    hi = x.to(torch.bfloat16)
    return torch.stack([hi, (x - hi.float()).to(torch.bfloat16)], 0).contiguous()


qkv3 = _split(torch.randn(16 * 1370, 3 * 1024, device=DEV))
ao3 = torch.zeros(2, 16 * 1370, 1024, device=DEV, dtype=torch.bfloat16)
under_load("attention bf16x3 B16 N1370 H16", lambda: hip.attention_x3(qkv3, ao3, 16, 1370, 16, 0.125))
A3, W3 = _split(torch.randn(M, 1024, device=DEV)), _split(torch.randn(3072, 1024, device=DEV) * 0.03)
o3 = torch.zeros(2, M, 3072, device=DEV, dtype=torch.bfloat16)
under_load("gemm bf16x3 QKV M21920 N3072 K1024", lambda: hip.gemm_x3(A3, W3, M, 3072, 1024, o3, zero))
del qkv3, ao3, A3, W3, o3

# the benchmark step itself (UFM-Base, 8 pairs, two micro-batch streams): what share of the power cap does a whole step draw?
import ufm_amd  # noqa: E402
from ufm_amd.modules import init_weights_  # noqa: E402

model = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_base_config()).eval()
init_weights_(model, 0)
model = model.to(DEV)
g = torch.Generator().manual_seed(1)
src = torch.randint(0, 256, (8, 518, 518, 3), dtype=torch.uint8, generator=g).to(DEV)
tgt = torch.randint(0, 256, (8, 518, 518, 3), dtype=torch.uint8, generator=g).to(DEV)
under_load("bench step: UFM-Base B=8 518^2 fast (2 streams)", lambda: model.predict_correspondences_batched(src, tgt), 4.0, 4)
model.engine().micro_batches = 1
under_load("bench step, one stream", lambda: model.predict_correspondences_batched(src, tgt), 4.0, 4)
