#!/usr/bin/env python3
"""Same-box A/B of ufm_attention_bf16 (pre-scaled q: the persistent kernel) between the in-tree library and another build of it
(tools/lab/bin/libufm_hip_old.so): interleaved rounds in one process, random data (cdna_hip_programming.md rules 24, 25)."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
libs = {"new": C.CDLL(os.path.join(ROOT, "ufm_amd", "libufm_hip.so")), "old": C.CDLL(os.path.join(ROOT, "tools", "lab", "bin", "libufm_hip_old.so"))}
for l in libs.values():
    l.ufm_attention_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
    l.ufm_attention_bf16.restype = C.c_int
st = torch.cuda.current_stream().cuda_stream
for name, b, n, h in (("enc", 16, 1370, 16), ("info", 8, 2738, 12), ("enc_mb", 8, 1370, 16), ("info_mb", 4, 2738, 12)):
    qkv = torch.randn(b * n, 3 * h * 64, device="cuda").bfloat16()
    qkv[:, : h * 64] = (qkv[:, : h * 64].float() * (0.125 * 1.4426950408889634)).bfloat16()
    outs = {k: torch.zeros(b * n, h * 64, device="cuda", dtype=torch.bfloat16) for k in libs}
    times = {k: [] for k in libs}
    for rnd in range(9):
        for k, l in libs.items():
            run = lambda: l.ufm_attention_bf16(qkv.data_ptr(), outs[k].data_ptr(), b, n, h, 0.0, st)
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record(); torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / 10)
    fl = 4.0 * b * h * n * n * 64
    msg = f"{name}: "
    for k in libs:
        t = sorted(times[k]); med = t[len(t) // 2]
        msg += f" {k} {med * 1e3:.1f} us {fl / med / 1e9:.0f} TF (min {t[0] * 1e3:.1f})"
    print(msg, "bitwise equal:", torch.equal(outs["new"], outs["old"]), flush=True)
