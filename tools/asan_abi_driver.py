#!/usr/bin/env python3
"""Drives the host side of the C ABI in the AddressSanitizer + UBSan build (ufm_amd/libufm_hip_asan.so, `make -C ufm_amd/csrc asan`):
every entry point's argument validation with hostile arguments, the error-string plumbing, and -- past validation -- the GEMM /
conv dispatch cost models and the launch bookkeeping (on a box without a GPU the launch itself fails cleanly with
UFM_ERR_LAUNCH; device code is never run here).  Run under LD_PRELOAD of the sanitizer runtime (tests/test_abi_cpu.py does);
any sanitizer report aborts the process with a non-zero exit code."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ufm_amd import hip  # noqa: E402  (signature table only; its own loader is not used)

lib = C.CDLL(os.path.join(ROOT, "ufm_amd", "libufm_hip_asan.so"))
for name, argtypes in hip.SIGNATURES.items():
    fn = getattr(lib, name)
    fn.argtypes, fn.restype = argtypes, C.c_int
lib.ufm_last_error.restype = C.c_char_p
lib.ufm_abi_version.restype = C.c_int
assert lib.ufm_abi_version() == hip.ABI_VERSION
P = C.c_void_p(4096)  # a non-null, 16-byte aligned fake device pointer: never dereferenced on the host
calls = 0


def expect(rc_ok, rc, what):
    global calls
    calls += 1
    msg = lib.ufm_last_error()
    assert rc in rc_ok, (what, rc, msg)
    assert rc == 0 or (msg and len(msg) > 0), what


# 1. every entry point with all-zero / null arguments: must be rejected (or be a setter), never crash
for name, argtypes in hip.SIGNATURES.items():
    args = []
    for t in argtypes:
        if t is C.c_void_p:
            args.append(None)
        elif t in (C.c_int, C.c_int64):
            args.append(0)
        elif t is C.c_float:
            args.append(0.0)
        else:  # pointer to a small host array
            args.append(None)
    rc = getattr(lib, name)(*args)
    expect((0, -1, -2), rc, name)
# 2. shape contracts of the hot entry points
expect((-1,), lib.ufm_gemm_bf16(P, 96, P, 96, 4, 128, 96, None, 0, None, None, 0, 0, P, 0, 128, 0, None), "gemm K % 64")
expect((-1,), lib.ufm_gemm_bf16(P, 64, P, 64, 4, 100, 64, None, 0, None, None, 0, 0, P, 0, 128, 0, None), "gemm N % 128")
expect((-1,), lib.ufm_gemm_bf16(P, 64, P, 64, 1 << 30, 1 << 20, 64, None, 0, None, None, 0, 0, P, 0, 1 << 20, 0, None), "gemm too large")
expect((-1,), lib.ufm_attention_bf16(None, None, 1, 1, 1, 0.125, None), "attention null")
expect((-1,), lib.ufm_attention_bf16_strided(P, 64, 16, P, P, 64, 128, P, 64, 64, 1, 64, 128, 1, 0.0, None), "strided batch rows")
expect((-1,), lib.ufm_attention_bf16_strided(P, 60, 64, P, P, 64, 128, P, 64, 64, 1, 64, 128, 1, 0.0, None), "strided ldq")
# ufm_cross_attention_bf16 by `scale` (include/ufm_hip.h): scale > 0 keeps the looser output contract (ldo % 4, 8-byte aligned out);
# scale == 0 (q pre-scaled: the LDS-DMA kernel) needs ldo % 8 and a 16-byte aligned out; scale < 0 is an argument error
P8 = C.c_void_p(P.value + 8)
expect((-1,), lib.ufm_cross_attention_bf16(P, 64, P, P, 64, P, 68, 1, 64, 64, 1, 0.0, None), "cross scale=0 ldo % 8")
expect((-1,), lib.ufm_cross_attention_bf16(P, 64, P, P, 64, P8, 64, 1, 64, 64, 1, 0.0, None), "cross scale=0 out alignment")
expect((-1,), lib.ufm_cross_attention_bf16(P, 64, P, P, 64, P, 64, 1, 64, 64, 1, -0.125, None), "cross scale<0")
expect((-1,), lib.ufm_cross_attention_bf16(P, 64, P, P, 64, P, 66, 1, 64, 64, 1, 0.125, None), "cross scale>0 ldo % 4")
expect((-1,), lib.ufm_debug_set_gemm_stamps(P, 0), "stamps buffer without rows")
expect((0,), lib.ufm_debug_set_gemm_stamps(None, 0), "stamps off")
expect((-1,), lib.ufm_debug_set_conv_stamps(None, 5), "conv stamps rows without buffer")
expect((0,), lib.ufm_debug_set_conv_stamps(None, 0), "conv stamps off")
# ufm_hint_concurrent_stream: a 32-entry table of opaque handles (no device call): null refused, flag / re-flag / remove, overflow reported
lib.ufm_hint_concurrent_stream.argtypes = [C.c_void_p, C.c_int]
lib.ufm_hint_concurrent_stream.restype = C.c_int
expect((-1,), lib.ufm_hint_concurrent_stream(None, 1), "null stream cannot be flagged")
for h in range(1, 33):
    expect((0,), lib.ufm_hint_concurrent_stream(C.c_void_p(0x1000 * h), 1), "flag a stream")
expect((0,), lib.ufm_hint_concurrent_stream(C.c_void_p(0x1000), 1), "flag it again")
expect((-1,), lib.ufm_hint_concurrent_stream(C.c_void_p(0x999000), 1), "table full")
for h in range(1, 33):
    expect((0,), lib.ufm_hint_concurrent_stream(C.c_void_p(0x1000 * h), 0), "remove the flag")
expect((0,), lib.ufm_hint_concurrent_stream(C.c_void_p(0x999000), 0), "removing an unknown stream is not an error")
# round 6: flags are reference-counted per handle -- 0x1000 was taken twice and given back once, so it still owns a slot
for h in range(2, 33):
    expect((0,), lib.ufm_hint_concurrent_stream(C.c_void_p(0x1000 * h), 1), "31 other streams fit")
expect((-1,), lib.ufm_hint_concurrent_stream(C.c_void_p(0x999000), 1), "the twice-flagged handle still holds its slot")
expect((0,), lib.ufm_hint_concurrent_stream(C.c_void_p(0x1000), 0), "second holder gives it back")
expect((0,), lib.ufm_hint_concurrent_stream(C.c_void_p(0x999000), 1), "now the slot is free")
for h in list(range(2, 33)) + [0x999]:
    expect((0,), lib.ufm_hint_concurrent_stream(C.c_void_p(0x1000 * h), 0), "clean up")
# the lab flag words refuse bits outside their field tables (csrc/lab_flags.h)
expect((-1,), lib.ufm_debug_set_gemm_flags(1), "bit 0 belongs to no field")
expect((-1,), lib.ufm_debug_set_conv_variant(1 << 21), "bit 21 belongs to no field")
expect((0,), lib.ufm_debug_set_conv_variant(0), "conv variant off")
expect((-1,), lib.ufm_gather_rows_f32(P, 62, P, 4, 64, P, 64, None), "gather ld")
expect((-1,), lib.ufm_debug_set_gemm_variant(3), "variant")
expect((-1,), lib.ufm_debug_set_gemm_tile_rows(100), "tile rows")
# 3. past validation: the dispatch cost models over a sweep of shapes, every variant / tile-height override (a launch without a
#    GPU returns UFM_ERR_LAUNCH after the host code has run; with a GPU the fake pointers must not be used: skip there)
try:
    import torch

    has_gpu = torch.cuda.is_available()
except Exception:
    has_gpu = False
if not has_gpu:
    for variant in (0, 1, 4, 5, 6, 7):
        lib.ufm_debug_set_gemm_variant(variant)
        for rows in (0, 160, 192, 224, 256):
            lib.ufm_debug_set_gemm_tile_rows(rows)
            for M in (1, 127, 2738, 10952, 21920, 65536):
                for N, K in ((128, 64), (768, 768), (1024, 1024), (1024, 4096), (3072, 1024), (4096, 1024), (2304, 768)):
                    for out_dtype in (0, 1):
                        rc = lib.ufm_gemm_bf16(P, K, P, K, M, N, K, P, 1 if out_dtype else 0, P if not out_dtype else None, P if not out_dtype else None, N, 0, P, out_dtype, N, 0, None)
                        expect((0, -2), rc, ("gemm dispatch", variant, rows, M, N, K, out_dtype))
    lib.ufm_debug_set_gemm_variant(0)
    lib.ufm_debug_set_gemm_tile_rows(0)
    for B, H, Cin, Cout, k in ((1, 19, 256, 256, 3), (8, 148, 256, 256, 3), (8, 296, 256, 128, 3), (2, 74, 768, 256, 1), (1, 37, 32, 32, 3)):
        rc = lib.ufm_conv2d_nhwc_bf16x3(P, B, H, H, Cin, P, Cout, k, k, 1, k // 2, 0, P, 0, None, None, 0, P, None, P, 3, None)
        expect((0, -2), rc, ("conv dispatch", B, H, Cin, Cout, k))
    expect((0, -2), lib.ufm_attention_bf16(P, P, 2, 1370, 16, 0.0, None), "attention launch bookkeeping")
    expect((0, -2), lib.ufm_cross_attention_bf16(P, 64, P, P, 64, P8, 68, 1, 64, 64, 1, 0.125, None), "cross scale>0: ldo % 4 and an 8-byte aligned out are enough")
    for flags in (1 << 24, 2 << 24, 4 << 24, 8 << 24):  # the pair kernel's auto rules, flipped on
        lib.ufm_debug_set_gemm_flags(flags)
        for M in (2738, 10952, 21920):
            for N, K, od in ((2304, 768, 1), (768, 768, 0), (768, 3072, 0), (3072, 1024, 1), (3072, 768, 1)):
                rc = lib.ufm_gemm_bf16(P, K, P, K, M, N, K, P, 0, P, P if not od else None, N, 0, P, od, N, 0, None)
                expect((0, -2), rc, ("pair rules", flags, M, N, K, od))
    lib.ufm_debug_set_gemm_flags(0)
    expect((0, -2), lib.ufm_layernorm(P, 1024, None, 100, 1024, P, P, 1e-6, P, 1, 1024, None), "layernorm")
print(f"asan abi driver ok: {calls} calls, no sanitizer report")
