"""ctypes binding of ``libufm_hip.so`` (C ABI declared in ``include/ufm_hip.h``).

This is the whole "FFI": raw device pointers (``tensor.data_ptr()``), ints, floats and the
current HIP stream.  PyTorch is used only to own device memory and streams.  There is NO
fallback: if the shared library is missing or a call fails, a ``RuntimeError`` is raised.
"""

from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libufm_hip.so")

F32, BF16, BF16X2, BF16X2_IL = 0, 1, 2, 3
ABI_VERSION = 2  # include/ufm_hip.h UFM_ABI_VERSION: bumped whenever an argument changes meaning, so a stale .so fails to load
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2

_lib: Optional[C.CDLL] = None

_vp, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64
_fp3 = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)

# name -> argtypes; MUST mirror include/ufm_hip.h (tests/test_abi.py checks the symbol set)
SIGNATURES = {
    "ufm_patchify": [_vp, _i, _i, _i, _i, _i, _i, _fp3, _fp3, _vp, _i, _i, _vp],
    "ufm_resize_antialias": [_vp, _i, _i, _i, _i, _i, _fp3, _fp3, _vp, _i, _i, _vp, _vp],
    "ufm_gemm_bf16": [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp],
    "ufm_gemm_bf16_rope": [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp],
    "ufm_rope2d": [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    "ufm_cross_attention_bf16": [_vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _vp],
    "ufm_attention_bf16_strided": [_vp, _i, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _vp],
    "ufm_cross_attention_f32": [_vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _vp],
    "ufm_cross_attention_bf16x3": [_vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _vp],
    "ufm_hint_concurrent_stream": [_vp, _i],
    "ufm_debug_set_gemm_variant": [_i],
    "ufm_debug_set_gemm_flags": [_i],
    "ufm_debug_set_gemm_tile_rows": [_i],
    "ufm_debug_set_gemm_splitk": [_vp, _vp, C.c_longlong, _i],
    "ufm_debug_lab_field": [_i, _i, C.POINTER(C.c_char_p), _ip, _ip],
    "ufm_debug_set_gemm_stamps": [_vp, _i],
    "ufm_debug_set_conv_stamps": [_vp, _i],
    "ufm_debug_set_attn_variant": [_i],
    "ufm_debug_set_conv_variant": [_i],
    "ufm_debug_set_upsample_variant": [_i],
    "ufm_warp_bilinear": [_vp, _i, _i, _i, _vp, _i, _i, _vp, _i, _f, _vp, _vp],
    "ufm_dpt_tail_fused": [_vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "ufm_layernorm": [_vp, _i, _vp, _i, _i, _vp, _vp, _f, _vp, _i, _i, _vp],
    "ufm_layernorm_slice": [_vp, _i, _vp, _i, _i, _vp, _vp, _f, _vp, _i, _i, _i64, _vp],
    "ufm_add_layernorm": [_vp, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _f, _vp, _i, _i, _vp],
    "ufm_gather_rows_f32": [_vp, _i, _vp, _i, _i, _vp, _i, _vp],
    "ufm_fill_rows": [_vp, _i, _i, _i, _vp, _i, _vp],
    "ufm_add_rows": [_vp, _i, _vp, _i, _i, _vp, _i, _i, _i, _i, _vp],
    "ufm_attention_bf16": [_vp, _vp, _i, _i, _i, _f, _vp],
    "ufm_attention_f32": [_vp, _vp, _i, _i, _i, _f, _vp],
    "ufm_debug_attention_stamps": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "ufm_conv2d_nhwc_f32": [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp],
    "ufm_conv2d_nhwc_bf16x3": [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp],
    "ufm_conv2d_nhwc_bf16x3_grouped": [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, C.c_longlong, _vp],
    "ufm_gemm_bf16x3": [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp],
    "ufm_gemm_bf16x3_il": [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp],
    "ufm_attention_bf16x3": [_vp, _vp, _i, _i, _i, _f, _vp],
    "ufm_attention_bf16x3_il": [_vp, _vp, _i, _i, _i, _f, _vp],
    "ufm_upsample_bilinear_nhwc": [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp],
    "ufm_head_tail": [_vp, _i, _i, _i, _i, _vp, _vp, _i, _ip, _fp3, _fp3, _vp, _vp, _vp],
    "ufm_adaptor_covariance2d": [_vp, _i, _i, _vp, _vp, _vp, _vp],
    "ufm_adaptor_confidence": [_vp, _i64, _i, _f, _f, _vp, _vp],
    "ufm_unmap_flow": [_vp, _i, _i, _i, _ip, _ip, _ip, _i, _i, _vp, _vp, _vp],
    "ufm_unmap_channels": [_vp, _i, _i, _i, _i, _ip, _ip, _i, _i, _fp3, _vp, _vp, _vp],
    "ufm_refine": [_vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp],
    "ufm_image_to_nhwc": [_vp, _i, _i, _i, _i, _i, _fp3, _fp3, _vp, _i, _i, _vp],
    "ufm_maxpool2x2_nhwc": [_vp, _i, _i, _i, _i, _i, _vp, _vp],
    "ufm_resize_nearest_nhwc": [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp],
    "ufm_unet_combine": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "ufm_group_norm_nhwc": [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _vp],
    "ufm_fill_uv_nhwc": [_vp, _i, _i, _i, _i, _i, _i, _f, _vp],
    "ufm_resize_bilinear_nhwc": [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp],
    "ufm_pixel_shuffle_planar": [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "ufm_cast_f32_to_bf16": [_vp, _vp, _i64, _vp],
    "ufm_add_f32": [_vp, _vp, _vp, _i64, _vp],
}
SIGNATURES["ufm_conv_x3_register_interleaved_weights"] = [_vp, _vp]
PLAIN = {"ufm_conv_x3_splitk_ws_bytes": (C.c_longlong, [C.c_int] * 10), "ufm_group_norm_ws_floats": (C.c_int, [C.c_int, C.c_int, C.c_int]), "ufm_abi_version": (C.c_int, []), "ufm_last_error": (C.c_char_p, []), "ufm_built_arch": (C.c_char_p, [])}


def lib() -> C.CDLL:
    """Load the library (once).  Raises loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C ufm_amd/csrc`). "
                "ufm_amd has no CPU / PyTorch fallback."
            )
        l = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = argtypes
            fn.restype = C.c_int
        for name, (res, argtypes) in PLAIN.items():
            fn = getattr(l, name)
            fn.argtypes = argtypes
            fn.restype = res
        if l.ufm_abi_version() != ABI_VERSION:
            raise RuntimeError("libufm_hip.so ABI version mismatch")
        _lib = l
    return _lib


class KernelTimer:
    """Optional HIP-event bracket around every C-ABI launch (bench.py's per-kernel durations).
    Events are recorded on the stream the kernels are launched on (torch's current stream OF THE CALLING THREAD: the engine's two
    micro-batch workers each bracket their own launches on their own stream; `streams[i]` is the raw handle records[i] ran on)."""

    def __init__(self, concurrent: bool = False):
        # concurrent=False (default): the engine runs single-stream while the timer is set (launch durations do not overlap: the per-kernel
        # roofline leg); True: the engine keeps its micro-batch / head streams (the timed configuration; durations overlap across streams)
        self.concurrent = concurrent
        self.records = []  # (name, start_event, end_event, meta)
        self.streams = []  # raw stream handle of each record
        self._tl = threading.local()
        self._lock = threading.Lock()

    @property
    def _open(self):
        return getattr(self._tl, "open", None)

    @_open.setter
    def _open(self, v):
        self._tl.open = v

    def begin(self, name: str, meta=None):
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        self._open = (name, e0, meta)

    def end(self):
        name, e0, meta = self._open
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        with self._lock:
            self.records.append((name, e0, e1, meta))
            self.streams.append(_stream())

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1, meta in self.records:
            d = out.setdefault(name, dict(ms=0.0, launches=0, metas=[]))
            d["ms"] += e0.elapsed_time(e1)
            d["launches"] += 1
            d["metas"].append(meta)
        return out


TIMER: Optional[KernelTimer] = None


def timer_serialises() -> bool:
    """True while a KernelTimer that wants non-overlapping launches is set: the engine then runs one stream."""
    return TIMER is not None and not TIMER.concurrent

_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)
if _raw_stream is None or _cur_device is None:  # an older / newer torch without the accessors: the public (slower) route
    _raw_stream = lambda dev: torch.cuda.current_stream(dev).cuda_stream  # noqa: E731
    _cur_device = torch.cuda.current_device


def _check(rc: int, name: str, family: Optional[str] = None) -> None:
    """`family`: the KernelTimer name this entry point is booked under when it differs from the entry point's own (the _il forms)."""
    if TIMER is not None and TIMER._open is not None and TIMER._open[0] == (family or name):
        TIMER.end()
        TIMER._open = None
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {lib().ufm_last_error().decode()}")


def _t(name: str, meta=None) -> None:
    if TIMER is not None:
        TIMER.begin(name, meta)


def _stream() -> int:
    """Raw handle of torch's current HIP stream on the current device.  torch.cuda.current_stream() builds a Stream object through
    four layers of device-index helpers -- 8 us x ~700 launches = 40 % of the host time of a one-pair forward (cProfile,
    tools/lab/host_profile_b1.py); the C accessor below returns the same handle (it is what torch's own compiled graphs use)."""
    return _raw_stream(_cur_device())


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    assert t.is_cuda, "ufm_amd kernels need device tensors"
    return t.data_ptr()


def _f3(v: Sequence[float]):
    return (C.c_float * len(v))(*[float(x) for x in v])


def _i4(v: Sequence[int]):
    return (C.c_int32 * len(v))(*[int(x) for x in v])


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(t.dtype)


# ----------------------------------------------------------------------------- wrappers
def patchify(img: torch.Tensor, layout: int, B: int, H: int, W: int, patch: int, scale3, shift3, out: torch.Tensor, kpad: int):
    in_dtype = 0 if img.dtype == torch.uint8 else 1
    assert img.is_contiguous() and out.is_contiguous()
    _check(lib().ufm_patchify(_p(img), in_dtype, layout, B, H, W, patch, _f3(scale3), _f3(shift3), _p(out), _dt(out), kpad, _stream()), "ufm_patchify")


def resize_antialias(img: torch.Tensor, layout: int, B: int, H: int, W: int, scale3, shift3, out: torch.Tensor, Ho: int, Wo: int, tmp: torch.Tensor):
    in_dtype = 0 if img.dtype == torch.uint8 else 1
    assert img.is_contiguous() and tmp.numel() >= B * 3 * H * Wo
    _check(lib().ufm_resize_antialias(_p(img), in_dtype, layout, B, H, W, _f3(scale3), _f3(shift3), _p(out), Ho, Wo, _p(tmp), _stream()), "ufm_resize_antialias")


def hint_concurrent_stream(stream: "torch.cuda.Stream", on: bool = True) -> bool:
    """Tell the library that ``stream`` runs side by side with other streams of the caller (ufm_hint_concurrent_stream): launches on it then
    choose tile heights for CU time, not for their own latency.  A hint: returns False instead of raising when the table is full."""
    return hint_concurrent_stream_handle(stream.cuda_stream, on)


def hint_concurrent_stream_handle(handle: int, on: bool = True) -> bool:
    """The same by raw hipStream_t value (an Engine's finalizer un-flags its streams after the torch objects are gone).  The library counts
    references per handle: every successful on=True must be paired with one on=False."""
    return lib().ufm_hint_concurrent_stream(C.c_void_p(handle), 1 if on else 0) == 0


def last_error() -> str:
    return lib().ufm_last_error().decode(errors="replace")


def gemm_bf16(A, W, M, N, K, out, *, bias=None, act=ACT_NONE, gamma=None, res=None, ldres=0, res_row_mod=0, lda=None, ldw=None, ldo=None, out_row_group=0, rope=None):
    """rope = (cos_table, sin_table, mod, cols): RoPE-2D fused into the epilogue on output columns [0, cols) (ufm_gemm_bf16_rope)."""
    if rope is not None:
        _t("ufm_gemm_bf16", (2.0 * M * N * K, f"M{M} N{N} K{K} bf16 out + RoPE"))
        _check(
            lib().ufm_gemm_bf16_rope(_p(A), lda or K, _p(W), ldw or K, M, N, K, _p(bias), act, _p(gamma), _p(res), ldres or N, res_row_mod, _p(out), _dt(out), ldo or N, out_row_group,
                                     _p(rope[0]), _p(rope[1]), int(rope[2]), int(rope[3]), _stream()),
            "ufm_gemm_bf16_rope",
        )
        return
    _t("ufm_gemm_bf16", (2.0 * M * N * K, f"M{M} N{N} K{K} " + ("f32 += (read-modify-write)" if (out.dtype == torch.float32 and res is not None and res_row_mod == 0) else
                                                             "f32 out" if out.dtype == torch.float32 else "bf16 out" + (" GELU" if act == ACT_GELU else ""))))
    _check(
        lib().ufm_gemm_bf16(_p(A), lda or K, _p(W), ldw or K, M, N, K, _p(bias), act, _p(gamma), _p(res), ldres or N, res_row_mod, _p(out), _dt(out), ldo or N, out_row_group, _stream()),
        "ufm_gemm_bf16",
    )


def layernorm(x, ldx, row_index, rows_out, D, weight, bias, eps, out, ldo=None, split=False, out_plane=None, interleaved=False):
    """split=True: `out` is a (2, rows_out, D) bf16 tensor in the UFM_BF16X2 format.  out_plane (elements): `out` is a row
    slice of a larger split buffer whose planes are that far apart (ufm_layernorm_slice).  interleaved=True (with split): `out` is
    (rows_out, D // 32, 2, 32) bf16 -- UFM_BF16X2_IL, the operand format of gemm_x3_il."""
    _t("ufm_layernorm", rows_out * D * (4.0 + (4 if split else out.element_size())))
    if interleaved:
        assert split and out_plane is None and out.dtype == torch.bfloat16
        _check(lib().ufm_layernorm(_p(x), ldx, _p(row_index), rows_out, D, _p(weight), _p(bias), eps, _p(out), BF16X2_IL, ldo or D, _stream()), "ufm_layernorm")
        return
    if out_plane is not None:
        _check(lib().ufm_layernorm_slice(_p(x), ldx, _p(row_index), rows_out, D, _p(weight), _p(bias), eps, _p(out), BF16X2 if split else _dt(out), ldo or D, int(out_plane), _stream()), "ufm_layernorm_slice")
        return
    _check(lib().ufm_layernorm(_p(x), ldx, _p(row_index), rows_out, D, _p(weight), _p(bias), eps, _p(out), BF16X2 if split else _dt(out), ldo or D, _stream()), "ufm_layernorm")


def gather_rows(x, ldx, row_index, rows, D, out, ldo=None):
    """out[r] = x[row_index[r]] (fp32 rows)."""
    _t("ufm_gather_rows_f32", rows * D * 8.0)
    _check(lib().ufm_gather_rows_f32(_p(x), ldx, _p(row_index), rows, D, _p(out), ldo or D, _stream()), "ufm_gather_rows_f32")


def add_layernorm(x, ldx, branch, gamma, rows, D, weight, bias, eps, out, ldo=None, split=False):
    """x[r] += gamma * branch[r] (x updated in place), then out = LayerNorm(x).  branch: bf16 (rows, D)."""
    _t("ufm_add_layernorm", rows * D * (4.0 + 2.0 + 4.0 + (4 if split else out.element_size())))
    assert branch.dtype == torch.bfloat16 and x.dtype == torch.float32
    _check(lib().ufm_add_layernorm(_p(x), ldx, _p(branch), branch.shape[-1], _p(gamma), rows, D, _p(weight), _p(bias), eps, _p(out), BF16X2 if split else _dt(out), ldo or D, _stream()), "ufm_add_layernorm")


def fill_rows(out, ldo, n_groups, group_stride_rows, src, D):
    _check(lib().ufm_fill_rows(_p(out), ldo, n_groups, group_stride_rows, _p(src), D, _stream()), "ufm_fill_rows")


def add_rows(a, lda, tab, tab_mod, out, ldo, out_row_group, rows, D, ldtab=None):
    _check(lib().ufm_add_rows(_p(a), lda, _p(tab), ldtab or D, tab_mod, _p(out), ldo, out_row_group, rows, D, _stream()), "ufm_add_rows")


def attention(qkv, out, B, N, H, scale):
    if qkv.dtype == torch.bfloat16:
        _t("ufm_attention_bf16", 4.0 * B * H * N * N * 64)
        _check(lib().ufm_attention_bf16(_p(qkv), _p(out), B, N, H, scale, _stream()), "ufm_attention_bf16")
    else:
        _t("ufm_attention_f32", 4.0 * B * H * N * N * 64)
        _check(lib().ufm_attention_f32(_p(qkv), _p(out), B, N, H, scale, _stream()), "ufm_attention_f32")


def gemm_x3(A, W, M, N, K, out, zero_page, *, bias=None, act=ACT_NONE, gamma=None, res=None):
    """Linear on the split format (numerics "precise"): A (2, M, K) / W (2, N, K) bf16 planes; out (2, M, N) bf16 planes,
    or fp32 (M, N) with the optional fp32 residual `res` (may be `out`)."""
    split_out = out.dtype == torch.bfloat16
    assert A.dtype == torch.bfloat16 and W.dtype == torch.bfloat16 and (split_out or out.dtype == torch.float32)
    _t("ufm_gemm_bf16x3", (2.0 * M * N * K, f"M{M} N{N} K{K} " + ("split out" + (" GELU" if act == ACT_GELU else "") if split_out else "f32 += (read-modify-write)" if res is not None else "f32 out")))
    _check(lib().ufm_gemm_bf16x3(_p(A), _p(W), M, N, K, _p(bias), act, _p(gamma), _p(res), _p(out), BF16X2 if split_out else F32, _p(zero_page), _stream()), "ufm_gemm_bf16x3")


def gemm_x3_il(A, W, M, N, K, out, zero_page, *, bias=None, act=ACT_NONE, gamma=None, res=None):
    """gemm_x3 on INTERLEAVED split operands: A (M, K // 32, 2, 32), W (N, K // 32, 2, 32) bf16 (interleave_split()); outputs as gemm_x3, or -- a
    4-D bf16 `out` (M, N // 32, 2, 32) -- interleaved too."""
    split_out = out.dtype == torch.bfloat16
    out_il = split_out and out.dim() == 4
    assert A.dtype == torch.bfloat16 and W.dtype == torch.bfloat16 and (split_out or out.dtype == torch.float32)
    _t("ufm_gemm_bf16x3", (2.0 * M * N * K, f"M{M} N{N} K{K} " + ("split out" + (" GELU" if act == ACT_GELU else "") if split_out else "f32 += (read-modify-write)" if res is not None else "f32 out") + " [il]"))
    _check(lib().ufm_gemm_bf16x3_il(_p(A), _p(W), M, N, K, _p(bias), act, _p(gamma), _p(res), _p(out), (BF16X2_IL if out_il else BF16X2) if split_out else F32, _p(zero_page), _stream()), "ufm_gemm_bf16x3_il", "ufm_gemm_bf16x3")


def interleave_split(planes: torch.Tensor) -> torch.Tensor:
    """(2, rows, C) split planes -> (rows, C // 32, 2, 32): the UFM_BF16X2_IL layout (a pack-time / test-time re-layout, plain torch)."""
    two, rows, C = planes.shape
    assert two == 2 and C % 32 == 0
    return planes.view(2, rows, C // 32, 32).permute(1, 2, 0, 3).contiguous()


def attention_x3(qkv, out, B, N, H, scale, out_interleaved=False):
    """Attention on the split format: qkv (2, B*N, 3*H*64), out (2, B*N, H*64) bf16 planes -- or, out_interleaved, (B*N, H*2, 2, 32): UFM_BF16X2_IL."""
    _t("ufm_attention_bf16x3", 4.0 * B * H * N * N * 64)
    if out_interleaved:
        _check(lib().ufm_attention_bf16x3_il(_p(qkv), _p(out), B, N, H, scale, _stream()), "ufm_attention_bf16x3_il", "ufm_attention_bf16x3")
        return
    _check(lib().ufm_attention_bf16x3(_p(qkv), _p(out), B, N, H, scale, _stream()), "ufm_attention_bf16x3")


def rope2d(x, rows, ld, col0, ncols, cos_table, sin_table, mod):
    """In-place RoPE-2D on columns [col0, col0 + ncols) of x: fp32 / bf16 (rows, ld), or split planes (2, rows, ld)."""
    fmt = BF16X2 if (x.dtype == torch.bfloat16 and x.dim() == 3) else _dt(x)
    _check(lib().ufm_rope2d(_p(x), fmt, rows, ld, col0, ncols, _p(cos_table), _p(sin_table), mod, _stream()), "ufm_rope2d")


def cross_attention(q, ldq, k, v, ldkv, out, ldo, B, Nq, Nk, H, scale, fmt):
    """Two-source attention; q / k / v / out may be column views of wider buffers (pass their data pointers through
    tensors whose storage offset is already applied).  fmt: F32, BF16 or BF16X2."""
    name = {F32: "ufm_cross_attention_f32", BF16: "ufm_cross_attention_bf16", BF16X2: "ufm_cross_attention_bf16x3"}[fmt]
    _t({F32: "ufm_attention_f32", BF16: "ufm_attention_bf16", BF16X2: "ufm_attention_bf16x3"}[fmt], 4.0 * B * H * Nq * Nk * 64)
    rc = getattr(lib(), name)(_p(q), ldq, _p(k), _p(v), ldkv, _p(out), ldo, B, Nq, Nk, H, scale, _stream())
    if TIMER is not None and TIMER._open is not None:
        TIMER.end()
        TIMER._open = None
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {lib().ufm_last_error().decode()}")


def attention_strided(q, ldq, q_batch_rows, k, v, ldkv, kv_batch_rows, out, ldo, out_batch_rows, B, Nq, Nk, H, scale=0.0):
    """bf16 two-source attention with explicit per-batch-item row strides (ufm_attention_bf16_strided); scale = 0: q pre-scaled,
    the persistent LDS-DMA kernel."""
    _t("ufm_attention_bf16", 4.0 * B * H * Nq * Nk * 64)
    _check(lib().ufm_attention_bf16_strided(_p(q), ldq, q_batch_rows, _p(k), _p(v), ldkv, kv_batch_rows, _p(out), ldo, out_batch_rows, B, Nq, Nk, H, scale, _stream()),
           "ufm_attention_bf16_strided")


def conv2d(x, B, H, W, Cin, weight, Cout, KH, KW, stride, pad, out, zero_page, *, relu_in=False, bias=None, act=ACT_NONE, gamma=None, res1=None, res2=None, shuffle=0, replicate=False):
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    _t("ufm_conv2d_nhwc_f32", 2.0 * B * Ho * Wo * Cout * KH * KW * Cin)
    _check(
        lib().ufm_conv2d_nhwc_f32(_p(x), B, H, W, Cin, _p(weight), Cout, KH, KW, stride, pad, int(relu_in) | (2 if replicate else 0), _p(bias), act, _p(gamma), _p(res1), _p(res2), shuffle, _p(out), 0, _p(zero_page), _stream()),
        "ufm_conv2d_nhwc_f32",
    )


def conv2d_x3(x, B, H, W, Cin, weight, Cout, KH, KW, stride, pad, out, zero_page, *, relu_in=False, bias=None, act=ACT_NONE, res1=None, res2=None, shuffle=0, out_relu=None, passes=3, replicate=False,
              groups=1, in_shared=False, splitk_ws=None):
    """bf16x3 split-precision conv; x / weight / res / out are (2, ...) bf16 tensors (UFM_BF16X2).
    passes=1: the hi planes only (a plain bf16 convolution with fp32 accumulation), same operand and output format.
    groups > 1: that many convolutions of identical geometry in one launch (ufm_conv2d_nhwc_bf16x3_grouped); B is per group."""
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    if groups > 1 or splitk_ws is not None:
        _t("ufm_conv2d_nhwc_bf16x3", (2.0 * groups * B * Ho * Wo * Cout * KH * KW * Cin, (f"G{groups} " if groups > 1 else "") + f"B{B} {H}x{W} {Cin}->{Cout} k{KH}" + (f" s{stride}" if stride != 1 else "") + (f" shuffle{shuffle}" if shuffle else "")))
        _check(
            lib().ufm_conv2d_nhwc_bf16x3_grouped(_p(x), groups, int(in_shared), B, H, W, Cin, _p(weight), Cout, KH, KW, stride, pad, int(relu_in) | (2 if replicate else 0), _p(bias), act, _p(res1), _p(res2),
                                                 shuffle, _p(out), _p(out_relu), _p(zero_page), passes, _p(splitk_ws), splitk_ws.numel() * splitk_ws.element_size() if splitk_ws is not None else 0, _stream()),
            "ufm_conv2d_nhwc_bf16x3",  # (the timer's family name; the entry point is ufm_conv2d_nhwc_bf16x3_grouped)
        )
        return
    _t("ufm_conv2d_nhwc_bf16x3" if passes == 3 else "ufm_conv2d_nhwc_bf16x1",
       (2.0 * B * Ho * Wo * Cout * KH * KW * Cin, f"B{B} {H}x{W} {Cin}->{Cout} k{KH}" + (f" s{stride}" if stride != 1 else "") + (f" shuffle{shuffle}" if shuffle else "")))
    _check(
        lib().ufm_conv2d_nhwc_bf16x3(_p(x), B, H, W, Cin, _p(weight), Cout, KH, KW, stride, pad, int(relu_in) | (2 if replicate else 0), _p(bias), act, _p(res1), _p(res2), shuffle, _p(out), _p(out_relu), _p(zero_page), passes, _stream()),
        "ufm_conv2d_nhwc_bf16x3",
    )


def upsample_bilinear(x, B, H, W, C, out, Ho, Wo, crop_h=0, crop_w=0):
    split = x.dtype == torch.bfloat16
    _t("ufm_upsample_bilinear_nhwc", 4.0 * B * C * (H * W + (crop_h or Ho) * (crop_w or Wo)))
    _check(lib().ufm_upsample_bilinear_nhwc(_p(x), BF16X2 if split else F32, B, H, W, C, _p(out), Ho, Wo, crop_h, crop_w, _stream()), "ufm_upsample_bilinear_nhwc")


def head_tail(x, P, HW, Cin, w, b, Cout, kinds, a, d, out, out_logits=None):
    _check(lib().ufm_head_tail(_p(x), BF16X2 if x.dtype == torch.bfloat16 else F32, P, HW, Cin, _p(w), _p(b), Cout, _i4(kinds), _f3(a), _f3(d), _p(out), _p(out_logits), _stream()), "ufm_head_tail")


def dpt_tail_fused(x, B, h, w, Cin, w2, b2, Cmid, H, W, wt, bt, Ct, kinds, a, d, out, out_logits=None, in_plane=0):
    """upsample -> conv3x3 + ReLU -> conv1x1 -> adaptor in one kernel (split-bf16 input and 3x3 weights).
    in_plane: elements between the hi and lo plane of x when x is one head's slice of a stacked buffer (0 = dense)."""
    _t("ufm_dpt_tail_fused", 2.0 * B * H * W * Cmid * 9 * Cin)
    _check(lib().ufm_dpt_tail_fused(_p(x), B, h, w, Cin, _p(w2), _p(b2), Cmid, H, W, _p(wt), _p(bt), Ct, _i4(kinds), _f3(a), _f3(d), _p(out), _p(out_logits), int(in_plane), _stream()), "ufm_dpt_tail_fused")


def warp_bilinear(target, flow, out, mask=None, mask_mode=0, fill=0.0):
    """target: (Ht, Wt, 3) uint8 or fp32; flow: (2, H, W) fp32; out: (H, W, 3) fp32; mask: (H, W) fp32 or None."""
    Ht, Wt = target.shape[:2]
    H, W = flow.shape[1:]
    _check(lib().ufm_warp_bilinear(_p(target), 0 if target.dtype == torch.uint8 else 1, Ht, Wt, _p(flow), H, W, _p(mask), mask_mode, float(fill), _p(out), _stream()), "ufm_warp_bilinear")


def unmap_flow(flow, B, h, w, rep0, src0, src1, H0, W0, out, valid=None):
    _check(lib().ufm_unmap_flow(_p(flow), B, h, w, _i4(rep0), _i4(src0), _i4(src1), H0, W0, _p(out), _p(valid), _stream()), "ufm_unmap_flow")


def unmap_channels(chan, B, Cc, h, w, rep0, src0, H0, W0, out, valid=None, chan_scale=None):
    cs = _f3(chan_scale) if chan_scale is not None else None
    _check(lib().ufm_unmap_channels(_p(chan), B, Cc, h, w, _i4(rep0), _i4(src0), H0, W0, cs, _p(out), _p(valid), _stream()), "ufm_unmap_channels")


def adaptor_covariance2d(raw, B, HW, cov, inv_cov, log_det):
    _check(lib().ufm_adaptor_covariance2d(_p(raw), B, HW, _p(cov), _p(inv_cov), _p(log_det), _stream()), "ufm_adaptor_covariance2d")


def adaptor_confidence(raw, kind: int, vmin: float, vmax: float, out):
    _check(lib().ufm_adaptor_confidence(_p(raw), raw.numel(), kind, float(vmin), float(vmax), _p(out), _stream()), "ufm_adaptor_confidence")


def refine(flow, feat, B, Cc, H, W, P, temperature, bias, residual, log_softmax=None):
    _t("ufm_refine", 4.0 * B * H * W * (2 * Cc + 4 + (P * P if log_softmax is not None else 0)))  # both feature maps once, flow, residual, log-softmax
    _check(lib().ufm_refine(_p(flow), _p(feat), B, Cc, H, W, P, temperature, _p(bias), _p(residual), _p(log_softmax), _stream()), "ufm_refine")


def _fmt(t: torch.Tensor) -> int:
    """NHWC activation format of a head/UNet buffer: fp32, or the split format (leading dim 2, bf16)."""
    return BF16X2 if t.dtype == torch.bfloat16 else F32


def image_to_nhwc(img, layout, B, H, W, scale3, shift3, out, Cpad):
    in_dtype = 0 if img.dtype == torch.uint8 else 1
    _t("ufm_image_to_nhwc", float(B) * H * W * (3 * img.element_size() + 4 * Cpad))
    _check(lib().ufm_image_to_nhwc(_p(img), in_dtype, layout, B, H, W, _f3(scale3), _f3(shift3), _p(out), _fmt(out), Cpad, _stream()), "ufm_image_to_nhwc")


def maxpool2x2(x, B, H, W, Cc, out):
    _t("ufm_maxpool2x2_nhwc", 4.0 * B * Cc * (H * W + (H // 2) * (W // 2)))
    _check(lib().ufm_maxpool2x2_nhwc(_p(x), _fmt(x), B, H, W, Cc, _p(out), _stream()), "ufm_maxpool2x2_nhwc")


def resize_nearest(x, B, H, W, Cc, out, Ho, Wo, ldc, c_off):
    _t("ufm_resize_nearest_nhwc", 8.0 * B * Ho * Wo * Cc)
    _check(lib().ufm_resize_nearest_nhwc(_p(x), _fmt(x), B, H, W, Cc, _p(out), Ho, Wo, ldc, c_off, _stream()), "ufm_resize_nearest_nhwc")


def unet_combine(cls, unet, N, HW, ldu, w1, b1, w2, b2, method, out):
    _t("ufm_unet_combine", 4.0 * N * HW * 48)
    _check(lib().ufm_unet_combine(_p(cls), _p(unet), _fmt(unet), N, HW, ldu, _p(w1), _p(b1), _p(w2), _p(b2), method, _p(out), _stream()), "ufm_unet_combine")


def group_norm(x, B, HW, Cc, groups, weight, bias, eps, relu, out, ws):
    """nn.GroupNorm(groups, C) (+ ReLU) on an NHWC map (fp32 or split); ws: fp32 workspace of group_norm_ws_floats()."""
    _t("ufm_group_norm_nhwc", 16.0 * B * HW * Cc)
    _check(lib().ufm_group_norm_nhwc(_p(x), _fmt(x), B, HW, Cc, Cc, groups, _p(weight), _p(bias), eps, int(relu), _p(out), _p(ws), _stream()), "ufm_group_norm_nhwc")


def group_norm_ws_floats(B, HW, groups) -> int:
    return int(lib().ufm_group_norm_ws_floats(B, HW, groups))


def fill_uv(out, B, H, W, ldc, c_off, aspect_ratio):
    _check(lib().ufm_fill_uv_nhwc(_p(out), _fmt(out), B, H, W, ldc, c_off, float(aspect_ratio), _stream()), "ufm_fill_uv_nhwc")


def resize_bilinear(x, B, H, W, Cc, ldi, out, Ho, Wo, ldc, c_off):
    """F.interpolate(bilinear, align_corners=False) into channels [c_off, c_off + C) of a wider NHWC map."""
    _t("ufm_resize_bilinear_nhwc", 4.0 * B * Cc * (H * W + Ho * Wo))
    _check(lib().ufm_resize_bilinear_nhwc(_p(x), _fmt(x), B, H, W, Cc, ldi, _p(out), Ho, Wo, ldc, c_off, _stream()), "ufm_resize_bilinear_nhwc")


def pixel_shuffle_planar(x, B, gh, gw, Cc, p, out, split=False):
    """split=True: `x` is a (2, rows, C*p*p) bf16 tensor in the UFM_BF16X2 format."""
    _t("ufm_pixel_shuffle_planar", 8.0 * B * gh * gw * p * p * Cc)
    _check(lib().ufm_pixel_shuffle_planar(_p(x), BF16X2 if split else F32, B, gh, gw, Cc, p, _p(out), _stream()), "ufm_pixel_shuffle_planar")


def cast_bf16(x, out):
    _check(lib().ufm_cast_f32_to_bf16(_p(x), _p(out), x.numel(), _stream()), "ufm_cast_f32_to_bf16")


def add_f32(a, b, out):
    _check(lib().ufm_add_f32(_p(a), _p(b), _p(out), a.numel(), _stream()), "ufm_add_f32")
