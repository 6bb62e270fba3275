"""HIP execution engine for the UFM forward pass.

Owns the packed device weights and the activation workspace and issues the C-ABI kernels of
``libufm_hip.so`` on the current HIP stream, in the order of the reference's ``forward``
(``uniflowmatch/models/ufm.py:562-662`` / ``:843-1009``).  Everything between "uint8 / float
image on the device" and "flow / covisibility at network resolution" happens here; no torch
compute op is used (torch only allocates buffers).

Data layout in HBM
  * tokens-major activations ``[rows][channels]``; encoder rows are ``(image, token)`` with the
    2B images ordered [view-1 batch | view-2 batch] (ufm.py:308) and token 0 = cls; info-sharing
    rows are ``(pair, view, patch)`` so one pair's 2*Np joint tokens are contiguous.
  * fp32 residual stream; bf16 GEMM/attention operands in numerics mode "fast" (the reference's
    own GPU policy: bf16 autocast trunk, base.py:273), fp32 everywhere in mode "parity".
  * DPT heads run in fp32 in both modes (ufm.py:635) on NHWC maps == token rows, so no transposes.
  * feature pyramids are gathered by the LayerNorm kernel through row-index tables (drop cls,
    pick view 1) -- the reference's ``.float().contiguous()`` copies (ufm.py:602-630) do not exist.

Numerics modes
  "fast"    bf16 MFMA operands, fp32 accumulate, fp32 residual / LayerNorm / softmax statistics.
  "precise" every contraction of the trunk AND the heads in bf16x3 split precision (hi*hi + hi*lo + lo*hi on the bf16
            matrix cores, activations carried as (hi, lo) bf16 planes, fp32 residual stream / LayerNorm / softmax
            statistics): tracks the fp32 CPU oracle to <= 1e-3 px at a third of the bf16 MFMA rate.
  "parity"  exact-fp32 MFMA for every contraction; tracks the fp32 CPU oracle to <= 1e-3 px.
"""

from __future__ import annotations

import threading
import warnings
import weakref
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F  # only for the one-off pos-embed interpolation at pack time
from torch import nn

from . import hip
from .modules import IMAGE_NORMALIZATION

KPAD = 640  # 3*14*14 = 588 patch columns padded to a multiple of the GEMM K-step


def _f32(t: torch.Tensor, dev) -> torch.Tensor:
    return t.detach().to(device=dev, dtype=torch.float32).contiguous()


def _split_bf16(w: torch.Tensor) -> torch.Tensor:
    """fp32 -> UFM_BF16X2: stack(hi = bf16(w), lo = bf16(w - hi))."""
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    return torch.stack([hi, lo], dim=0).contiguous()


class _Lin:
    """Packed nn.Linear: weight in the mode's operand dtype, bias fp32."""

    def __init__(self, lin: nn.Linear, dev, wdt, split: bool = False, interleave: bool = False):
        self.n, self.k = lin.weight.shape
        self.il = False
        if split:  # UFM_BF16X2 (2, N, K) for ufm_gemm_bf16x3 -- or, round 6, UFM_BF16X2_IL (N, K / 32, 2, 32) for ufm_gemm_bf16x3_il
            self.w = _split_bf16(lin.weight.detach().to(device=dev, dtype=torch.float32))
            if interleave:
                self.w, self.il = hip.interleave_split(self.w), True
        else:
            self.w = lin.weight.detach().to(device=dev, dtype=wdt).contiguous()
        self.b = _f32(lin.bias, dev) if lin.bias is not None else None


LOG2E = 1.4426950408889634

# Bumped whenever a Parameter / sub-module / buffer is registered on a module that belongs to a TRACKED model (one an Engine has walked;
# torch's global registration hooks also fire for ``module.weight = nn.Parameter(..)``): the engines cache their model's Parameter
# objects and re-walk the module tree only when their model's epoch moved.  Round 6: the epoch used to be process-global, so
# every nn.Module construction anywhere -- a second model, user code -- made every engine re-walk its tree (0.7-1.7 ms of host time) on
# its next call; now a registration counts only when the touched module is in a tracked tree (_TRACKED: module -> the epoch cells of the
# engines whose model contains it).  PARAM_EPOCH stays as the count of ALL registrations (tests, diagnostics).
PARAM_EPOCH = [0]
_TRACKED: "weakref.WeakKeyDictionary[nn.Module, Any]" = weakref.WeakKeyDictionary()  # module -> weakref.WeakSet of _Epoch cells


class _Epoch:
    """One engine's 'my model's tree may have changed' counter (weak-referenced from _TRACKED)."""

    __slots__ = ("n", "__weakref__")

    def __init__(self):
        self.n = 0


def _on_parameter_registration(module, name, param):  # noqa: ARG001
    PARAM_EPOCH[0] += 1
    cells = _TRACKED.get(module)
    if cells:
        for c in cells:
            c.n += 1


def _track_tree(model: nn.Module, cell: _Epoch) -> None:
    for mod in model.modules():
        s_ = _TRACKED.get(mod)
        if s_ is None:
            s_ = _TRACKED[mod] = weakref.WeakSet()
        s_.add(cell)


# The parameter hook alone misses a whole SUB-MODULE swapped in (``model.head1 = other_head``, ``seq[i] = copy.deepcopy(mod)``,
# ``add_module``): torch fires the module-registration hook for those, and the buffer hook for ``register_buffer`` / buffer assignment.
# (``del m.bias`` and direct writes to ``m._parameters[...]`` fire no hook at all: like edits through ``p.data`` they are invisible to
# any guard short of walking the tree on every call.)
torch.nn.modules.module.register_module_parameter_registration_hook(_on_parameter_registration)
torch.nn.modules.module.register_module_module_registration_hook(_on_parameter_registration)
torch.nn.modules.module.register_module_buffer_registration_hook(_on_parameter_registration)


def _il_ok(*lins: nn.Linear) -> bool:
    """ufm_gemm_bf16x3_il's contract for every Linear of a block: N % 256 == 0 (the 8-phase tile), K % 32 == 0, K >= 64."""
    return all(l.weight.shape[0] % 256 == 0 and l.weight.shape[1] % 32 == 0 and l.weight.shape[1] >= 64 for l in lins)


class _Blk:
    def __init__(self, blk, dev, wdt, split: bool = False):
        self.n1w, self.n1b = _f32(blk.norm1.weight, dev), _f32(blk.norm1.bias, dev)
        self.n2w, self.n2b = _f32(blk.norm2.weight, dev), _f32(blk.norm2.bias, dev)
        # numerics "precise" (split): the four Linears of a block on INTERLEAVED split operands where their shapes allow it (round 6:
        # every LDS-DMA row of the bf16x3 loop a whole 128-byte line, -7...-16 % per launch; profiles/r06/gemm_x3_il_ab.log) -- LayerNorm,
        # the attention kernel and the fc1 epilogue write that format (Engine._blocks)
        il = self.il = bool(split) and _il_ok(blk.attn.qkv, blk.attn.proj, blk.mlp.fc1, blk.mlp.fc2) and blk.norm1.weight.shape[0] % 256 == 0
        self.qkv, self.proj = _Lin(blk.attn.qkv, dev, wdt, split, il), _Lin(blk.attn.proj, dev, wdt, split, il)
        self.fc1, self.fc2 = _Lin(blk.mlp.fc1, dev, wdt, split, il), _Lin(blk.mlp.fc2, dev, wdt, split, il)
        self.ls1 = _f32(blk.ls1.gamma, dev) if hasattr(blk.ls1, "gamma") else None
        self.ls2 = _f32(blk.ls2.gamma, dev) if hasattr(blk.ls2, "gamma") else None
        # fast mode: the QKV epilogue scales the Q columns by softmax_scale*log2(e) BEFORE the bf16 rounding,
        # so the attention kernel works in the log2 domain with no per-score multiply and no extra rounding
        d = self.qkv.n // 3
        self.qscale = None
        # (round 6: the interleaved split path of "precise" pre-scales too -- the split store takes (hi, lo) of q * scale * log2(e), and
        #  ufm_attention_bf16x3_il with scale = 0 exponentiates the scores as they are: one VALU instruction per score less in an issue-bound kernel)
        if (wdt == torch.bfloat16 and not split) or il:
            self.qscale = torch.cat([torch.full((d,), 0.125 * LOG2E), torch.ones(2 * d)]).to(device=dev, dtype=torch.float32)


def _cat_linear(*lins: nn.Linear) -> nn.Linear:
    """Several Linears over the same input as one (rows of the weights stacked): projk | projv -> one K|V GEMM."""
    out = nn.Linear(lins[0].in_features, sum(l.out_features for l in lins), bias=lins[0].bias is not None)
    with torch.no_grad():
        out.weight.copy_(torch.cat([l.weight.detach().float().cpu() for l in lins], dim=0))
        if out.bias is not None:
            out.bias.copy_(torch.cat([l.bias.detach().float().cpu() for l in lins], dim=0))
    return out


class _XBlk(_Blk):
    """Packed CrossBlockParams: the self-attention half is a _Blk (norm1/attn/ls1, norm3 -> n2, mlp, ls3 -> ls2)."""

    def __init__(self, blk, dev, wdt, split: bool = False):
        self.il = False  # (the cross-attention variant keeps the planar split format)
        self.n1w, self.n1b = _f32(blk.norm1.weight, dev), _f32(blk.norm1.bias, dev)
        self.n2w, self.n2b = _f32(blk.norm3.weight, dev), _f32(blk.norm3.bias, dev)   # the MLP's norm
        self.qkv, self.proj = _Lin(blk.attn.qkv, dev, wdt, split), _Lin(blk.attn.proj, dev, wdt, split)
        self.fc1, self.fc2 = _Lin(blk.mlp.fc1, dev, wdt, split), _Lin(blk.mlp.fc2, dev, wdt, split)
        g = lambda m: _f32(m.gamma, dev) if hasattr(m, "gamma") else None  # noqa: E731
        self.ls1, self.ls2 = g(blk.ls1), g(blk.ls3)
        self.lsx = g(blk.ls2)                                                        # the cross-attention's LayerScale
        d = self.qkv.n // 3
        self.qscale = None
        if wdt == torch.bfloat16 and not split:
            self.qscale = torch.cat([torch.full((d,), 0.125 * LOG2E), torch.ones(2 * d)]).to(device=dev, dtype=torch.float32)
        self.nyw, self.nyb = _f32(blk.norm_y.weight, dev), _f32(blk.norm_y.bias, dev)
        self.nxw, self.nxb = _f32(blk.norm2.weight, dev), _f32(blk.norm2.bias, dev)   # the cross-attention's query norm
        ca = blk.cross_attn
        self.q = _Lin(ca.projq, dev, wdt, split)
        # fast mode: the cross-attention's q projection pre-scales by softmax_scale * log2(e) too, so the two-source attention
        # runs on the persistent LDS-DMA kernel (ufm_cross_attention_bf16 with scale == 0)
        self.qxscale = torch.full((d,), 0.125 * LOG2E, device=dev, dtype=torch.float32) if (wdt == torch.bfloat16 and not split) else None
        self.kv = _Lin(_cat_linear(ca.projk, ca.projv), dev, wdt, split)
        self.projx = _Lin(ca.proj, dev, wdt, split)


class _Conv:
    """Conv2d packed [Cout][KH][KW][Cin]; ConvTranspose2d(k == s) packed [(kh,kw,co)][Cin].
    split=True stores the weight pre-split in the UFM_BF16X2 format for the bf16x3 kernel."""

    def __init__(self, conv: nn.Module, dev, split: bool = False, cin_pad: int = 0, cout_pad: int = 0):
        w = conv.weight.detach().to(device=dev, dtype=torch.float32)
        self.b = _f32(conv.bias, dev) if conv.bias is not None else None
        if isinstance(conv, nn.ConvTranspose2d) and cin_pad > w.shape[0]:  # weight (cin, co, kh, kw): zero rows for the pad channels
            w = torch.cat([w, torch.zeros((cin_pad - w.shape[0],) + tuple(w.shape[1:]), device=dev, dtype=torch.float32)], dim=0)
        elif cin_pad or cout_pad:  # zero-pad to the kernels' channel multiples (UNet: 3 input channels, 16 output channels)
            assert isinstance(conv, nn.Conv2d)
            co, ci = w.shape[0], w.shape[1]
            wp = torch.zeros((max(co, cout_pad), max(ci, cin_pad)) + tuple(w.shape[2:]), device=dev, dtype=torch.float32)
            wp[:co, :ci] = w
            w = wp
            if self.b is not None and cout_pad > co:
                self.b = torch.cat([self.b, torch.zeros(cout_pad - co, device=dev)])
        if isinstance(conv, nn.ConvTranspose2d):
            cin, co, kh, kw = w.shape
            assert kh == kw == conv.stride[0] == conv.stride[1] and conv.padding == (0, 0)
            self.shuffle, self.cin, self.cout, self.k, self.stride, self.pad = kh, cin, kh * kw * co, 1, 1, 0
            self.w = w.permute(2, 3, 1, 0).reshape(kh * kw * co, cin).contiguous()
        else:
            co, cin, kh, kw = w.shape
            self.shuffle, self.cin, self.cout, self.k, self.stride, self.pad = 0, cin, co, kh, conv.stride[0], conv.padding[0]
            self.w = w.permute(0, 2, 3, 1).contiguous()
        if split:
            self.w = _split_bf16(self.w)
            # round 6: the layers the row-window halo kernel serves (3x3 / stride 1 / pad 1 on the 256-cout 8-phase tile) also keep their weights
            # INTERLEAVED per 32-channel chunk, registered against the planar tensor's address: the kernel then stages W as whole 128-byte lines
            # (-2...-3 % per launch on top of the halo tile, profiles/r06/conv_halo_ab.log).  The entry leaves the library's table with this object.
            if self.k == 3 and self.stride == 1 and self.pad == 1 and not self.shuffle and self.cout % 256 == 0 and self.cin % 32 == 0:
                self.w_il = hip.interleave_split(self.w.view(2, self.cout, 9 * self.cin))
                if hip.lib().ufm_conv_x3_register_interleaved_weights(self.w.data_ptr(), self.w_il.data_ptr()) == 0:
                    weakref.finalize(self, _unregister_wil, self.w.data_ptr())
                else:
                    self.w_il = None  # table full: the planar staging (same results)


def _unregister_wil(planar_ptr: int) -> None:
    try:
        hip.lib().ufm_conv_x3_register_interleaved_weights(planar_ptr, None)
    except Exception:  # interpreter shutdown
        pass


def _lin_as_conv(lin: nn.Linear, dev) -> _Conv:
    """nn.Linear as the 1x1 convolution it is, packed for the bf16x3 kernel."""
    conv = nn.Conv2d(lin.in_features, lin.out_features, kernel_size=1, bias=lin.bias is not None)
    with torch.no_grad():
        conv.weight.copy_(lin.weight.detach().float().cpu()[:, :, None, None])
        if lin.bias is not None:
            conv.bias.copy_(lin.bias.detach().float().cpu())
    return _Conv(conv, dev, split=True)


class _Head:
    def __init__(self, head: nn.Sequential, dev, split: bool = False):
        _C = lambda m, d: _Conv(m, d, split)  # noqa: E731
        feat, proc = head[0][0], head[0][1]
        self.feature_dim, self.layer_dims = feat.feature_dim, feat.layer_dims
        self.hooks = feat.hooks
        self.act = [[_C(m, dev) for m in seq] for seq in feat.act_postprocess]
        self.rn = [_C(getattr(feat.scratch, f"layer{i + 1}_rn"), dev) for i in range(4)]
        self.fuse = []
        for i in range(4):
            f = getattr(feat.scratch, f"refinenet{i + 1}")
            self.fuse.append(
                dict(
                    out=_C(f.out_conv, dev),
                    r1=(_C(f.resConfUnit1.conv1, dev), _C(f.resConfUnit1.conv2, dev)),
                    r2=(_C(f.resConfUnit2.conv1, dev), _C(f.resConfUnit2.conv2, dev)),
                )
            )
        self.p_conv1 = _C(proc.conv1, dev)
        self.p_conv2a = _C(proc.conv2[0], dev)
        last = proc.conv2[2]
        self.tail_w = _f32(last.weight.reshape(last.weight.shape[0], -1), dev)
        self.tail_b = _f32(last.bias, dev)
        self.tail_cout, self.tail_cin = last.weight.shape[0], last.weight.shape[1]
        self.adaptors = list(head[1].adaptors)
        self.kinds = [k for a in self.adaptors for k in a.kinds]
        self.scale = [s for a in self.adaptors for s in a.scale]
        self.shift = [s for a in self.adaptors for s in a.shift]


class _ConvG:
    """The same layer of several heads stacked for ufm_conv2d_nhwc_bf16x3_grouped: weights (2, G, Cout, ...), bias (G, Cout)."""

    def __init__(self, convs: List[_Conv]):
        c0 = convs[0]
        for c in convs[1:]:
            assert (c.cin, c.cout, c.k, c.stride, c.pad, c.shuffle) == (c0.cin, c0.cout, c0.k, c0.stride, c0.pad, c0.shuffle) and c.w.shape == c0.w.shape
            assert (c.b is None) == (c0.b is None)
        self.cin, self.cout, self.k, self.stride, self.pad, self.shuffle = c0.cin, c0.cout, c0.k, c0.stride, c0.pad, c0.shuffle
        self.groups = len(convs)
        self.w = torch.stack([c.w for c in convs], dim=1).contiguous()
        self.b = torch.stack([c.b for c in convs], dim=0).contiguous() if c0.b is not None else None


class _HeadG:
    """Several DPT heads of identical layer shapes (the flow head and the covisibility head: ufm.py:553-556) packed for one
    grouped launch per layer; only their last stage (3x3 128 -> 32, 1x1 -> output channels, adaptors) differs and stays per head."""

    def __init__(self, heads: List[_Head]):
        h0 = heads[0]
        self.heads = heads
        self.groups = len(heads)
        self.feature_dim, self.layer_dims, self.hooks = h0.feature_dim, h0.layer_dims, h0.hooks
        self.act = [[_ConvG([h.act[i][j] for h in heads]) for j in range(len(h0.act[i]))] for i in range(4)]
        self.rn = [_ConvG([h.rn[i] for h in heads]) for i in range(4)]
        self.fuse = [dict(out=_ConvG([h.fuse[i]["out"] for h in heads]),
                          r1=(_ConvG([h.fuse[i]["r1"][0] for h in heads]), _ConvG([h.fuse[i]["r1"][1] for h in heads])),
                          r2=(_ConvG([h.fuse[i]["r2"][0] for h in heads]), _ConvG([h.fuse[i]["r2"][1] for h in heads]))) for i in range(4)]
        self.p_conv1 = _ConvG([h.p_conv1 for h in heads])

    @staticmethod
    def compatible(heads) -> bool:
        if len(heads) < 2 or not all(isinstance(h, _Head) for h in heads):
            return False
        h0 = heads[0]
        sig = lambda c: (c.cin, c.cout, c.k, c.stride, c.pad, c.shuffle, tuple(c.w.shape), c.w.dtype, c.b is None)  # noqa: E731
        def layers(h):
            out = [sig(c) for seq in h.act for c in seq] + [sig(c) for c in h.rn] + [sig(h.p_conv1)]
            for f in h.fuse:
                out += [sig(f["out"]), sig(f["r1"][0]), sig(f["r1"][1]), sig(f["r2"][0]), sig(f["r2"][1])]
            return out
        return all(h.hooks == h0.hooks and h.layer_dims == h0.layer_dims and h.feature_dim == h0.feature_dim and layers(h) == layers(h0) for h in heads[1:]) \
            and h0.p_conv1.w.dtype == torch.bfloat16


class _MoGeHead:
    """Packed MoGeConvParams (modules.py): every convolution as a _Conv, GroupNorm affine vectors in fp32."""

    def __init__(self, head: nn.Sequential, dev, split: bool = False):
        feat = head[0]
        pad32 = lambda c: (c + 31) // 32 * 32  # noqa: E731
        self.dim_proj = feat.dim_proj
        self.projects = [_Conv(c, dev, split) for c in feat.projects]
        self.stages = []
        for blk in feat.upsample_blocks:
            ct, c3 = blk[0][0], blk[0][1]
            res = []
            for rb in list(blk)[1:]:
                gn1, conv_a, gn2, conv_b = rb.layers[0], rb.layers[2], rb.layers[3], rb.layers[5]
                res.append(dict(
                    gn1=(gn1.num_groups, _f32(gn1.weight, dev), _f32(gn1.bias, dev), gn1.eps), a=_Conv(conv_a, dev, split),
                    gn2=(gn2.num_groups, _f32(gn2.weight, dev), _f32(gn2.bias, dev), gn2.eps), b=_Conv(conv_b, dev, split),
                    skip=_Conv(rb.skip_connection, dev, split) if isinstance(rb.skip_connection, nn.Conv2d) else None,
                ))
            self.stages.append(dict(cin=ct.in_channels - 2, ldc=pad32(ct.in_channels), ct=_Conv(ct, dev, split, cin_pad=pad32(ct.in_channels)), c3=_Conv(c3, dev, split), res=res))
        ob = feat.output_block[0]
        self.out_cin, self.out_ldc = ob[0].in_channels - 2, pad32(ob[0].in_channels)
        self.out_conv = _Conv(ob[0], dev, split, cin_pad=self.out_ldc)
        last = ob[2]
        self.tail_w = _f32(last.weight.reshape(last.weight.shape[0], -1), dev)
        self.tail_b = _f32(last.bias, dev)
        self.tail_cout, self.tail_cin = last.weight.shape[0], last.weight.shape[1]
        self.adaptors = list(head[1].adaptors)
        self.kinds = [k for a in self.adaptors for k in a.kinds]
        self.scale = [s for a in self.adaptors for s in a.scale]
        self.shift = [s for a in self.adaptors for s in a.shift]


# model -> {numerics: weakref to the _Pack}: one packed copy per (model, numerics), shared by its engines.  The ENGINES own the pack
# (strong reference, Engine._pack_ref); the cache only finds it again.  When the last engine of a numerics mode goes away
# (set_numerics() drops the eager engine; a GraphedPredictor is deleted) its 0.85-1.7 GB of packed weights are freed (the
# cache used to hold every mode a model had ever used until the model died).
_PACK_CACHE: "weakref.WeakKeyDictionary[nn.Module, Dict[str, Any]]" = weakref.WeakKeyDictionary()


class _Pack:
    """Immutable packed weights of one (model, device, numerics, parameter versions): `attrs` = the Engine attributes _pack set."""

    __slots__ = ("key", "attrs", "__weakref__")

    def __init__(self, key, attrs):
        self.key, self.attrs = key, attrs


_PER_ENGINE_STATE = frozenset(("_bufs", "_tables", "_packed_key", "_plist", "_plist_epoch", "_epoch", "_pack_ref", "_tls", "_streams", "_head_streams", "_level_streams", "_head_group",
                               "_stream_finalizer", "_flagged"))


def _unflag_streams(handles: List[int]) -> None:
    """weakref.finalize target of an Engine: give its micro-batch stream flags back (one reference per flag taken)."""
    while handles:
        try:
            hip.hint_concurrent_stream_handle(handles.pop(), False)
        except Exception:  # interpreter shutdown: the library may be gone
            return


_HINT_REFUSED = [False]


def _warn_hint_refused() -> None:
    """ufm_hint_concurrent_stream refused (its table of flagged streams is full): results are bitwise the same, but this engine's
    micro-batch launches keep the latency tile policy (-1...-2.8 % pairs/s in the two-stream pipeline).  Said once, not ignored."""
    if not _HINT_REFUSED[0]:
        _HINT_REFUSED[0] = True
        warnings.warn("ufm_amd: ufm_hint_concurrent_stream refused a micro-batch stream (" + hip.last_error() + "): the GEMM / convolution "
                      "tile policy falls back to per-launch latency on it; results are unchanged, throughput is 1-3 % lower. "
                      "Too many live engines with micro-batch streams?", RuntimeWarning, stacklevel=3)


class Engine:
    def __init__(self, model: nn.Module, numerics: str = "fast"):
        if numerics not in ("fast", "precise", "parity", "parity_x3heads"):
            raise ValueError("numerics must be 'fast', 'precise', 'parity' or 'parity_x3heads'")
        hip.lib()  # fail loudly right here if the extension is missing
        self.model = model
        self.numerics = numerics
        self.adt = torch.bfloat16 if numerics == "fast" else torch.float32  # GEMM/attention operand dtype
        self.trunk_x3 = numerics == "precise"  # transformer blocks on the split format (ufm_gemm_bf16x3 / ufm_attention_bf16x3)
        self.dev: Optional[torch.device] = None
        self._bufs: Dict[str, torch.Tensor] = {}
        self._tables: Dict[Any, Any] = {}
        self._packed_key = None
        self._pack_ref: Optional[_Pack] = None  # the shared packed weights this engine keeps alive (_PACK_CACHE holds them weakly)
        self._epoch = _Epoch()  # moves when a Parameter / sub-module / buffer is registered inside THIS model's tree
        self._tls = threading.local()  # .ns = workspace namespace of the micro-batch this thread is running
        # last_layer_view1: the LAST joint-attention block computes its view-1 rows only (queries, proj, MLP on half the rows; keys /
        # values still from both views) -- the reference decodes view 1 only (ufm.py:637-641) and nothing else reads that block's
        # view-2 rows (UFM-Refine's classification head does: off there).  Exact: bit-identical to the full block on the kept rows.
        self.last_layer_view1 = True
        self.mb_bounds = None  # lab: explicit micro-batch boundaries, e.g. [0, 5, 8]
        self.micro_batches = 2  # >1: the batch is split into micro-batches that run concurrently on separate HIP streams
        # joint_heads: the micro-batch streams run the TRUNK only (patchify .. info sharing: GEMM / attention launches whose
        # partial rounds and HBM-bound epilogues overlap across streams); their view-1 feature pyramids land in one
        # full-batch buffer and the DPT heads run ONCE on the whole batch (full-size convolution launches: a 148^2 RCU layer
        # at 8 images is 485 us against 2 x 270 us at 4 + 4), the two heads on two streams.  Same arithmetic per pair.
        self.joint_heads = True
        self._streams: List[torch.cuda.Stream] = []
        # raw handles of the micro-batch streams this engine flagged with ufm_hint_concurrent_stream: un-flagged when the engine dies
        # (the library's table is small and counts references per handle; torch hands stream handles out of a 32-entry pool, so a
        # flag left behind would re-colour some later engine's head or level stream)
        self._flagged: List[int] = []
        self._stream_finalizer = weakref.finalize(self, _unflag_streams, self._flagged)
        # "fast", optional: the residual stream's fp32 read-modify-write is done by the next LayerNorm launch
        # (ufm_add_layernorm) instead of the proj / fc2 GEMM epilogues.  Measured (tools/lab/ab_engine.py, interleaved, same
        # box, B=8 518^2): the GEMM family rises 0.327 -> 0.346 of peak but the step does not move (38.76 vs 38.91 ms with two
        # micro-batch streams, 41.43 vs 40.94 single-stream) -- the same bytes, issued by another kernel -- and the bf16
        # rounding of the branch costs accuracy (flow max-abs 0.035 -> 0.041 px).  Default: off (fp32 accumulator straight
        # into the stream).
        self.defer_residual = False
        # group_heads: DPT heads of identical layer shapes run every layer as ONE grouped launch (ufm_conv2d_nhwc_bf16x3_grouped) on a
        # head-major stacked batch -- twice the tiles per grid instead of two launch sequences on two streams; bit-identical.
        # OFF by default: measured same-box (round 4, profiles/r04/grouped_heads_ab.log) 215.0 vs 216.9 pairs/s at B = 8 and
        # 9.11-9.14 vs 9.01-9.15 ms graph replay at B = 1 -- the two streams already interleave one head's small-grid layers with
        # the other's large ones, which one serial chain of doubled grids does not
        self.group_heads = False
        # conv_splitk: deterministic split-K for the DPT heads' small-map long-K layers (19^2 / 37^2: latency-bound chains of 72-216
        # K-steps on grids of a few dozen tiles); the factor depends on the layer geometry only, so results stay batch-invariant.
        # OFF by default: same-box A/B (round 4, profiles/r04/conv_splitk_ab.log) one-pair graph replay 8.75-8.88 vs 8.86-8.96 ms
        # (-1 %), B = 8 217.7 vs 218.6 pairs/s (+0.4 % slower) -- the two head streams already hide those chains behind each
        # other -- and it changes the last bits of the heads' sums (9e-5 px); not worth a second set of results
        self.conv_splitk = False
        self.fused_tail = True  # ufm_dpt_tail_fused where the head has the UFM-Base tail shape (bit-identical to the unfused path)
        # DPT heads on separate HIP streams: None = automatic (yes for a single-stream forward -- one pair: 9.84 -> 8.97 ms
        # graph replay, the small-grid layers of one head fill the other's tails -- no inside a micro-batch worker, where the
        # other micro-batch already does that: 166 vs 168 pairs/s); True / False force it
        self.concurrent_heads: Optional[bool] = None
        self._head_streams: Dict[str, List[torch.cuda.Stream]] = {}
        # level_streams: the DPT heads' four level chains on side streams (graph branches under capture) when at most
        # `level_streams_max_images` images go through a head call.  OFF: slower at every batch size -- 8 pairs 37.8 vs 37.1 ms (round 3),
        # one pair 9.8 vs 9.0 ms graph replay, two pairs 15.2 vs 14.2 (round 4, profiles/r04/capture_nested_fork.log): eight streams
        # of latency-bound launches delay the critical fusion chain more than they shorten it.  Kept because it carries the fix of
        # the round-3 capture crash (forks start at the capture's origin stream; see _head) and the test that guards it.
        self.level_streams = False
        self.level_streams_max_images = 2
        self._level_streams: Dict[Any, List[torch.cuda.Stream]] = {}
        self._init_names = frozenset(self.__dict__) | {"_init_names"}  # everything _pack adds on top is packed weights / their metadata

    # ------------------------------------------------------------------ packing
    def _pack(self) -> None:
        m = self.model
        # This guard runs in front of EVERY forward: walking the module tree three times (m.parameters()) cost 0.7-1.7 ms of host time
        # per call -- part of the one-pair eager latency.  The Parameter objects are cached; the tree is walked again only after a
        # Parameter was (re-)registered on ANY module (PARAM_EPOCH: torch's global parameter-registration hook), i.e. when a Parameter
        # object may have been swapped into a sub-module; in-place edits, load_state_dict and .to() show in version / address below.
        plist = getattr(self, "_plist", None)
        if plist is None or self._plist_epoch != self._epoch.n:
            _track_tree(m, self._epoch)  # (a swapped-in sub-module's own children are tracked from here on)
            plist = self._plist = list(m.parameters())
            self._plist_epoch = self._epoch.n
        dev = plist[0].device
        if dev.type != "cuda":
            raise RuntimeError("ufm_amd runs on an AMD GPU only: move the model with .to('cuda') (no CPU fallback exists)")
        key = (str(dev), self.numerics, tuple((p._version, p.data_ptr()) for p in plist))
        if key == self._packed_key:
            return

        self.dev = dev
        self._tables.clear()
        self._bufs.clear()
        # The packed weights are immutable once built: every Engine of this model with the same (device, numerics, parameter
        # versions / storage) -- the eager engine and the private engine of each GraphedPredictor -- shares ONE copy (round 5; each
        # used to pack its own 0.85 GB).  Workspace, tables and streams stay per engine.  A graph captured on a shared pack keeps the
        # tensors alive through its engine's references even after the model's weights change and the cache entry is replaced.
        ref = _PACK_CACHE.setdefault(m, {}).get(self.numerics)
        cached = ref() if ref is not None else None
        if cached is not None and cached.key == key:
            self.__dict__.update(cached.attrs)
            self._pack_ref = cached
            self._head_group = None
            self._packed_key = key
            return
        enc = m.encoder.model
        self.P = enc.patch_size
        self.D = enc.embed_dim
        self.enc_heads = enc.num_heads
        self.enc_indices = list(m.encoder.indices)
        if self.D // self.enc_heads != 64 or m.info_sharing.dim // m.info_sharing.num_heads != 64:
            raise NotImplementedError("attention kernels are built for head_dim 64 (DINOv2 ViT-S/B/L and the UFM info-sharing)")
        wdt = self.adt
        pe_w = enc.patch_embed.proj.weight.detach().reshape(self.D, -1).float()
        assert pe_w.shape[1] == 3 * self.P * self.P <= KPAD
        w = torch.zeros(self.D, KPAD)
        w[:, : pe_w.shape[1]] = pe_w.cpu()
        self.pe_w = w.to(device=dev, dtype=wdt).contiguous()
        self.pe_b = _f32(enc.patch_embed.proj.bias, dev)
        self.enc_blocks = [_Blk(b, dev, wdt, self.trunk_x3) for b in enc.blocks]
        self.enc_norm = (_f32(enc.norm.weight, dev), _f32(enc.norm.bias, dev))
        info = m.info_sharing
        self.Di, self.info_heads, self.info_indices = info.dim, info.num_heads, list(info.indices)
        self.info_proj = _Lin(info.proj_embed, dev, wdt) if isinstance(info.proj_embed, nn.Linear) else None
        self.info_norm = (_f32(info.norm.weight, dev), _f32(info.norm.bias, dev))
        self.info_cross = hasattr(info, "multi_view_branches")  # the cross-attention variant (ufm.py:193, "cross_attention")
        if self.info_cross:
            self.info_branches = [[_XBlk(b, dev, wdt, self.trunk_x3) for b in branch] for branch in info.multi_view_branches]
            self.rope_freq = info.rope_freq
        else:
            self.info_blocks = [_Blk(b, dev, wdt, self.trunk_x3) for b in info.self_attention_blocks]
            if info.max_num_views < 2:
                raise ValueError("info sharing needs max_num_views >= 2")
        # DPT heads: exact-fp32 MFMA in "parity"; bf16x3 split precision (UFM_BF16X2 activations) in "fast"
        self.head_split = self.numerics in ("fast", "precise", "parity_x3heads")
        moge = not isinstance(m.head1[0], nn.Sequential)  # head_type "moge_conv" (ufm.py:266-267): one feature module, not (DPTFeature, processor)
        self._head_group = None  # _HeadG of self.heads, packed on first use
        self.heads = {"head1": (_MoGeHead if moge else _Head)(m.head1, dev, self.head_split)}
        if hasattr(m, "uncertainty_head"):  # built with its own head type (ufm.py:553-556 passes uncertainty_head_type)
            moge_u = not isinstance(m.uncertainty_head[0], nn.Sequential)
            self.heads["uncertainty_head"] = (_MoGeHead if moge_u else _Head)(m.uncertainty_head, dev, self.head_split)
        self.refine = hasattr(m, "classification_head")
        if self.refine:
            ch = m.classification_head
            # the classification head runs inside the fp32 island (ufm.py:921-965; the comment at :949 says "autocast", the
            # indentation says otherwise): exact-fp32 MFMA in "parity", the bf16x3 split form (a 1x1 convolution over the token
            # rows) wherever the DPT heads use it
            if self.head_split:
                self.cls_fc1, self.cls_fc2 = _lin_as_conv(ch.mlp.fc1, dev), _lin_as_conv(ch.mlp.fc2, dev)
                for c in (self.cls_fc1, self.cls_fc2):
                    c.n = c.cout  # (the Linear's output width, as _Lin names it)
            else:
                self.cls_fc1 = _Lin(ch.mlp.fc1, dev, torch.float32)
                self.cls_fc2 = _Lin(ch.mlp.fc2, dev, torch.float32)
            self.cls_out_dim = ch.output_dim
            self.cls_bias = _f32(m.classification_bias, dev)
            self.unet = None
            if getattr(m, "use_unet_feature", False):  # unet_encoder.py:26-71 + ufm.py:818-825
                u, sp = m.unet_feature, self.head_split
                dc = lambda d, first=False: (_Conv(d.conv[0], dev, sp, cin_pad=32 if first else 0), _Conv(d.conv[2], dev, sp))  # noqa: E731
                self.unet = dict(
                    downs=[dc(d, i == 0) for i, d in enumerate(u.downs)],
                    bottleneck=dc(u.bottleneck),
                    ups=[(_Conv(u.ups[k], dev, sp), dc(u.ups[k + 1])) for k in range(0, len(u.ups), 2)],
                    final=_Conv(u.final_conv, dev, sp, cout_pad=32),
                )
                if self.numerics == "fast":
                    # the reference runs the UNet OUTSIDE its fp32 island (ufm.py:915-917), i.e. under the bf16 autocast of
                    # base.py:273: plain bf16 operands, fp32 accumulation -- the hi planes of the split format, one MFMA pass
                    for c in [c for pair in self.unet["downs"] for c in pair] + list(self.unet["bottleneck"]) + \
                             [c for ct, pair in self.unet["ups"] for c in (ct,) + tuple(pair)] + [self.unet["final"]]:
                        c.passes = 1
                assert u.out_channels == 16 and self.cls_out_dim == 16, "the combine step is built for 16 + 16 feature channels (ufm.py:818-825)"
                self.unet_method = 0 if m.feature_combine_method == "conv" else 1
                self.unet_w1 = _f32(m.conv1.weight.reshape(m.conv1.weight.shape[0], -1), dev)
                self.unet_b1 = _f32(m.conv1.bias, dev)
                self.unet_w2 = _f32(m.conv2.weight.reshape(m.conv2.weight.shape[0], -1), dev)
                self.unet_b2 = _f32(m.conv2.bias, dev)
        self.zero = torch.zeros(256, device=dev, dtype=torch.float32)
        for d, what in ((self.D, "encoder dim"), (self.Di, "info-sharing dim")):
            if self.numerics == "fast" and (d % 128 != 0):
                raise NotImplementedError(f"{what}={d}: the bf16 GEMM tiles need channel counts that are multiples of 128")
            if d % 32 != 0:
                raise NotImplementedError(f"{what}={d} must be a multiple of 32")
        self._packed_key = key
        packed = {k: v for k, v in self.__dict__.items() if k not in _PER_ENGINE_STATE and k not in self._init_names}  # what _pack set
        self._pack_ref = _Pack(key, packed)
        _PACK_CACHE[m][self.numerics] = weakref.ref(self._pack_ref)

    # ------------------------------------------------------------------ small helpers
    def buf(self, name: str, shape: Tuple[int, ...], dtype=torch.float32) -> torch.Tensor:
        name = getattr(self._tls, "ns", "") + name
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, device=self.dev, dtype=dtype)
            self._bufs[name] = t
        return t

    def _pos_tables(self, H: int, W: int):
        """cls+pos[0] row and the (Np, D) patch pos-embed for this resolution (hub DINOv2 semantics:
        identity at the native grid, else bicubic with the +0.1 scale-factor offset).  One-off per
        resolution; the only place torch.nn.functional is touched."""
        key = ("pos", H, W)
        if key not in self._tables:
            enc = self.model.encoder.model
            pe = enc.pos_embed.detach().float().cpu()
            n_native = pe.shape[1] - 1
            gh, gw = H // self.P, W // self.P
            patch = pe[:, 1:]
            if not (gh * gw == n_native and gh == gw):
                mside = int(round(n_native**0.5))
                assert mside * mside == n_native
                off = enc.interpolate_offset
                patch = F.interpolate(
                    patch.reshape(1, mside, mside, -1).permute(0, 3, 1, 2),
                    mode="bicubic", antialias=False, scale_factor=((gh + off) / mside, (gw + off) / mside),
                )
                assert tuple(patch.shape[-2:]) == (gh, gw)
                patch = patch.permute(0, 2, 3, 1).reshape(1, gh * gw, -1)
            cls0 = (enc.cls_token.detach().float().cpu()[0, 0] + pe[0, 0]).contiguous()
            self._tables[key] = (cls0.to(self.dev), patch[0].contiguous().to(self.dev))
        return self._tables[key]

    def _index_tables(self, B: int, Np: int, symmetrized: bool = False):
        key = ("idx", B, Np, symmetrized)
        if key not in self._tables:
            N = Np + 1
            p = torch.arange(Np)
            b = torch.arange(B).view(B, 1)
            if symmetrized:
                # ufm.py:336-352: the encoder buffer holds B images = [img1[::2] | img2[::2]] = [a_0.. | b_0..]; pair 2i is
                # (a_i, b_i), pair 2i+1 is (b_i, a_i) (interleave, ufm.py:69-82)
                half = B // 2
                i, odd = torch.arange(B) // 2, torch.arange(B) % 2
                img_v1 = torch.where(odd == 0, i, half + i)  # encoder image that is view 1 of pair p
                img_v2 = torch.where(odd == 0, half + i, i)
            else:
                img_v1, img_v2 = torch.arange(B), B + torch.arange(B)  # images ordered [view-1 batch | view-2 batch] (ufm.py:308)
            v1 = (img_v1.view(B, 1) * N + 1 + p).reshape(-1)  # view-1 patch rows of the encoder buffer
            all_ = (torch.cat([img_v1, img_v2]).view(-1, 1) * N + 1 + p).reshape(-1)
            info = (torch.stack([img_v1, img_v2], dim=1).view(B, 2, 1) * N + 1 + p.view(1, 1, Np)).reshape(-1)  # (pair, view, patch) order
            iv1 = (b * 2 * Np + p).reshape(-1)
            iv2 = (b * 2 * Np + Np + p).reshape(-1)
            mk = lambda t: t.to(torch.int32).to(self.dev)  # noqa: E731
            self._tables[key] = dict(enc_v1=mk(v1), enc_all=mk(all_), enc_info=mk(info), info_v1=mk(iv1), info_v2=mk(iv2), info_all=mk(torch.cat([iv1, iv2])))
        return self._tables[key]

    def _view_pe_table(self, Np: int) -> torch.Tensor:
        key = ("vpe", Np)
        if key not in self._tables:
            t = self.model.info_sharing.view_pos_table.detach().float().cpu()
            self._tables[key] = torch.cat([t[0:1].expand(Np, -1), t[1:2].expand(Np, -1)], dim=0).contiguous().to(self.dev)
        return self._tables[key]

    def _rope_tables(self, gh: int, gw: int):
        """cos / signed-sin tables [gh*gw][64] of the CroCo RoPE-2D for one head (rope.hip): first 32 columns rotate with the
        token's y index, the last 32 with its x index; inside a half, pair (j, j ^ 16) shares theta = pos / freq^(2 (j % 16) / 32)."""
        key = ("rope", gh, gw, self.rope_freq)
        if key not in self._tables:
            ys, xs = torch.meshgrid(torch.arange(gh, dtype=torch.float64), torch.arange(gw, dtype=torch.float64), indexing="ij")
            pos = torch.stack([ys.reshape(-1), xs.reshape(-1)], dim=1)                       # (Np, 2) = (y, x)
            inv = 1.0 / (self.rope_freq ** (torch.arange(0, 32, 2, dtype=torch.float64) / 32))  # 16 frequencies
            cos, sin = torch.zeros(gh * gw, 64, dtype=torch.float64), torch.zeros(gh * gw, 64, dtype=torch.float64)
            for half in range(2):
                f = pos[:, half : half + 1] * inv                                               # (Np, 16)
                f = torch.cat([f, f], dim=1)                                                    # (Np, 32)
                sign = torch.cat([-torch.ones(16, dtype=torch.float64), torch.ones(16, dtype=torch.float64)])
                cos[:, half * 32 : half * 32 + 32] = f.cos().float().double()                   # (the oracle rounds cos / sin to fp32)
                sin[:, half * 32 : half * 32 + 32] = f.sin().float().double() * sign
            self._tables[key] = (cos.float().contiguous().to(self.dev), sin.float().contiguous().to(self.dev))
        return self._tables[key]

    def linear_rope(self, x, lin: _Lin, M: int, out, rope, rope_cols: int, gamma=None):
        """Projection whose first ``rope_cols`` output columns (q / k heads) get RoPE-2D: fused into the GEMM epilogue in
        "fast" (ufm_gemm_bf16_rope), the standalone in-place kernel after the Linear in the fp32 / split numerics."""
        if rope is None:
            return self.linear(x, lin, M, out, gamma=gamma)
        cos, sin, mod = rope
        if lin.w.dim() == 2 and lin.w.dtype == torch.bfloat16:
            hip.gemm_bf16(x, lin.w, M, lin.n, lin.k, out, bias=lin.b, gamma=gamma, rope=(cos, sin, mod, rope_cols))
        else:
            self.linear(x, lin, M, out, gamma=gamma)
            hip.rope2d(out, M, lin.n, 0, rope_cols, cos, sin, mod)

    def linear(self, x, lin: _Lin, M: int, out, *, act=hip.ACT_NONE, gamma=None, res=None, res_row_mod=0, out_row_group=0, lda=None):
        """out = epilogue(x @ W^T): bf16 MFMA GEMM in 'fast', the bf16x3 split form in 'precise', exact-fp32 MFMA (conv
        kernel as dense GEMM) in 'parity'."""
        if lin.il:  # UFM_BF16X2_IL weights: x is an (M, K / 32, 2, 32) interleaved split buffer; out planar split (3-D), interleaved (4-D) or fp32
            assert res_row_mod == 0 and out_row_group == 0 and lda is None and x.dim() == 4
            hip.gemm_x3_il(x, lin.w, M, lin.n, lin.k, out, self.zero, bias=lin.b, act=act, gamma=gamma, res=res)
        elif lin.w.dim() == 3:  # UFM_BF16X2 weights: x is a (2, M, K) split buffer, out split or the fp32 residual stream
            assert res_row_mod == 0 and out_row_group == 0 and lda is None
            hip.gemm_x3(x, lin.w, M, lin.n, lin.k, out, self.zero, bias=lin.b, act=act, gamma=gamma, res=res)
        elif lin.w.dtype == torch.bfloat16:
            hip.gemm_bf16(x, lin.w, M, lin.n, lin.k, out, bias=lin.b, act=act, gamma=gamma, res=res, res_row_mod=res_row_mod, out_row_group=out_row_group, lda=lda)
        else:
            assert res_row_mod == 0 and out_row_group == 0 and out.dtype == torch.float32
            hip.conv2d(x, 1, 1, M, lin.k, lin.w, lin.n, 1, 1, 1, 0, out, self.zero, bias=lin.b, act=act, gamma=gamma, res1=res)

    def conv(self, x, B, H, W, c: _Conv, out, *, relu_in=False, act=hip.ACT_NONE, res1=None, res2=None, out_relu=None, replicate=False, in_shared=False, ws: Optional[str] = None):
        """``c`` a _ConvG: the grouped launch (B = images per group; x / out / res hold groups * B images, group-major; ``in_shared``:
        x holds B images that every group reads).  ``ws``: name of the caller's split-K workspace (one per concurrent launch
        sequence): the small-map long-K layers then run the deterministic split-K form of ufm_conv2d_nhwc_bf16x3_grouped."""
        if c.w.dtype == torch.bfloat16:
            groups = getattr(c, "groups", 1)
            wsbuf = None
            if ws is not None and self.conv_splitk and not c.shuffle and getattr(c, "passes", 3) == 3:
                need = hip.lib().ufm_conv_x3_splitk_ws_bytes(groups, B, H, W, c.cin, c.cout, c.k, c.k, c.stride, c.pad)
                if need > 0:
                    wsbuf = self._splitk_ws(ws, need)
            hip.conv2d_x3(x, B, H, W, c.cin, c.w, c.cout, c.k, c.k, c.stride, c.pad, out, self.zero, relu_in=relu_in, bias=c.b, act=act, res1=res1, res2=res2, shuffle=c.shuffle, out_relu=out_relu,
                          passes=getattr(c, "passes", 3), replicate=replicate, groups=groups, in_shared=in_shared, splitk_ws=wsbuf)
        else:
            assert out_relu is None
            hip.conv2d(x, B, H, W, c.cin, c.w, c.cout, c.k, c.k, c.stride, c.pad, out, self.zero, relu_in=relu_in, bias=c.b, act=act, res1=res1, res2=res2, shuffle=c.shuffle, replicate=replicate)

    def _splitk_ws(self, name: str, nbytes: int) -> torch.Tensor:
        """Zero-filled split-K workspace ``name`` (grown to the largest layer that asked; the kernels leave its counters zero)."""
        name = getattr(self._tls, "ns", "") + name + "_splitk_ws"
        t = self._bufs.get(name)
        if t is None or t.numel() * 4 < nbytes:
            t = torch.zeros((nbytes + 3) // 4, device=self.dev, dtype=torch.float32)
            self._bufs[name] = t
        return t

    def hbuf(self, name: str, shape: Tuple[int, ...]) -> torch.Tensor:
        """Head activation buffer: fp32 [shape] or the split format (2, *shape) bf16."""
        if self.head_split:
            return self.buf(name, (2,) + tuple(shape), torch.bfloat16)
        return self.buf(name, shape)

    def level_ln(self, x, D, row_index, rows, w, b, name: str) -> torch.Tensor:
        """LayerNorm + row gather into a head-format feature level.  With joint heads the micro-batch's rows go straight
        into their slice of the full-batch level buffer (``_tls.joint``: level buffers by name, first row, total rows)."""
        joint = getattr(self._tls, "joint", None)
        if joint is not None:
            bufs, row0, rows_total = joint["levels"], joint["row0"], joint["rows_total"]
            full = bufs[name]  # (2, rows_total, D) split planes or (rows_total, D) fp32
            dst = full.narrow(-2, row0, rows)
            hip.layernorm(x, D, row_index, rows, D, w, b, 1e-6, dst, split=self.head_split, out_plane=rows_total * D)
            return dst
        t = self.hbuf(name, (rows, D))
        hip.layernorm(x, D, row_index, rows, D, w, b, 1e-6, t, split=self.head_split)
        return t

    # ------------------------------------------------------------------ transformer
    def _blocks(self, blocks: List[_Blk], x, Bseq: int, N: int, D: int, heads: int, on_block, needs_x=lambda i: True):
        """The transformer blocks on the fp32 residual stream ``x`` (updated in place).  ``needs_x(i)``: block i's output is
        read by ``on_block`` (or is the last one), so its residual update may not stay pending.

        "fast" with ``defer_residual``: proj / fc2 store their bf16 branch output (the cheap GEMM epilogue) and the NEXT
        LayerNorm launch applies ``x += gamma * branch`` in fp32 before normalising (ufm_add_layernorm) -- the reference's
        own arithmetic under bf16 autocast (the Linear returns bf16; LayerScale and the add run in fp32), with the residual
        stream's read-modify-write moved out of the GEMM epilogue, where all CUs hit HBM at once with the matrix cores idle."""
        M = Bseq * N
        x3 = self.trunk_x3
        tb = (lambda name, cols: self.buf(name + "_x2", (2, M, cols), torch.bfloat16)) if x3 else (lambda name, cols: self.buf(name, (M, cols), self.adt))
        il = x3 and all(w.il for w in blocks)  # the blocks' Linears read interleaved split operands: LayerNorm, attention and fc1 write them
        if il:
            ti = lambda name, cols: self.buf(name + "_il", (M, cols // 32, 2, 32), torch.bfloat16)  # noqa: E731
            xn, qkv, ao, hid = ti("xn", D), tb("qkv", 3 * D), ti("ao", D), ti("hid", blocks[0].fc1.n)  # (qkv stays planar: the attention kernel's K / V tiles)
        else:
            xn, qkv, ao, hid = tb("xn", D), tb("qkv", 3 * D), tb("ao", D), tb("hid", blocks[0].fc1.n)
        defer = self.numerics == "fast" and self.defer_residual
        br = self.buf("branch", (M, D), torch.bfloat16) if defer else None
        pending = False  # x still lacks gamma * br of the previous block's fc2
        pend_gamma = None
        for i, w in enumerate(blocks):
            if pending:
                hip.add_layernorm(x, D, br, pend_gamma, M, D, w.n1w, w.n1b, 1e-6, xn)
                pending = False
            else:
                hip.layernorm(x, D, None, M, D, w.n1w, w.n1b, 1e-6, xn, split=x3, interleaved=il)
            self.linear(xn, w.qkv, M, qkv, gamma=w.qscale)
            if x3:
                hip.attention_x3(qkv, ao, Bseq, N, heads, 0.0 if (il and w.qscale is not None) else 0.125, out_interleaved=il)
            else:
                hip.attention(qkv, ao, Bseq, N, heads, 0.0 if w.qscale is not None else 0.125)
            if defer:
                self.linear(ao, w.proj, M, br)
                hip.add_layernorm(x, D, br, w.ls1, M, D, w.n2w, w.n2b, 1e-6, xn)
            else:
                self.linear(ao, w.proj, M, x, gamma=w.ls1, res=x)
                hip.layernorm(x, D, None, M, D, w.n2w, w.n2b, 1e-6, xn, split=x3, interleaved=il)
            self.linear(xn, w.fc1, M, hid, act=hip.ACT_GELU)
            if defer and i + 1 < len(blocks) and not needs_x(i):
                self.linear(hid, w.fc2, M, br)
                pending, pend_gamma = True, w.ls2
            else:
                self.linear(hid, w.fc2, M, x, gamma=w.ls2, res=x)
            on_block(i, x)

    def _last_block_view1(self, w: _Blk, y, B: int, Np: int, Di: int, heads: int, v1_rows) -> torch.Tensor:
        """The last joint-attention block on its view-1 rows only ("fast" numerics): LayerNorm + QKV over all 2 Np tokens of a
        pair (keys / values of both views), then view-1 queries against all keys (ufm_attention_bf16_strided: q_batch_rows = 2 Np,
        Nq = Np), proj / MLP on the compact (B Np, Di) view-1 residual rows.  Returns that compact fp32 stream; ``y`` keeps the
        block's INPUT (its view-2 rows are never needed again: ufm.py:637-641 decodes view 1 only)."""
        M, M1 = B * 2 * Np, B * Np
        xn, qkv = self.buf("xn", (M, Di), self.adt), self.buf("qkv", (M, 3 * Di), self.adt)
        hip.layernorm(y, Di, None, M, Di, w.n1w, w.n1b, 1e-6, xn)
        self.linear(xn, w.qkv, M, qkv, gamma=w.qscale)
        ao1 = self.buf("ao_v1", (M1, Di), self.adt)
        hip.attention_strided(qkv[:, :Di], 3 * Di, 2 * Np, qkv[:, Di : 2 * Di], qkv[:, 2 * Di :], 3 * Di, 2 * Np, ao1, Di, Np, B, Np, 2 * Np, heads)
        y1 = self.buf("info_x_v1", (M1, Di))
        hip.gather_rows(y, Di, v1_rows, M1, Di, y1)
        self.linear(ao1, w.proj, M1, y1, gamma=w.ls1, res=y1)
        xn1 = self.buf("xn_v1", (M1, Di), self.adt)
        hip.layernorm(y1, Di, None, M1, Di, w.n2w, w.n2b, 1e-6, xn1)
        hid1 = self.buf("hid_v1", (M1, w.fc1.n), self.adt)
        self.linear(xn1, w.fc1, M1, hid1, act=hip.ACT_GELU)
        self.linear(hid1, w.fc2, M1, y1, gamma=w.ls2, res=y1)
        return y1

    def _view_major_tables(self, B: int, Np: int):
        key = ("vm", B, Np)
        if key not in self._tables:
            r = torch.arange(2 * B * Np, dtype=torch.int32)
            self._tables[key] = dict(v1=r[: B * Np].contiguous().to(self.dev), v2=r[B * Np :].contiguous().to(self.dev), all=r.to(self.dev))
        return self._tables[key]

    def _cross_blocks(self, y, B: int, Np: int, gh: int, gw: int, on_block) -> None:
        """The cross-attention info-sharing variant ([U] MultiViewCrossAttentionTransformerIFR, ufm.py:193): ``y`` holds the
        fp32 residual streams of both views, (view, pair, patch) rows.  Layer l updates BOTH views from the layer-(l-1)
        tokens: view v's block = self-attention over its own Np tokens, cross-attention of its queries to norm_y(the OTHER
        view's previous-layer tokens), MLP.  RoPE-2D (if configured) rotates q / k of both attentions: fused into the
        projection GEMM's epilogue in "fast".  Self-attention of the two views' B images runs through the same kernels as the
        encoder; the cross-attention through the two-source entry points (ufm_cross_attention_*)."""
        Di, heads, Mv = self.Di, self.info_heads, B * Np
        x3 = self.trunk_x3
        fmt = hip.BF16X2 if x3 else (hip.BF16 if self.adt == torch.bfloat16 else hip.F32)
        tb = (lambda name, rows, cols: self.buf(name + "_x2", (2, rows, cols), torch.bfloat16)) if x3 else (lambda name, rows, cols: self.buf(name, (rows, cols), self.adt))
        xn, qkv, ao, hid = tb("xc_xn", Mv, Di), tb("xc_qkv", Mv, 3 * Di), tb("xc_ao", Mv, Di), tb("xc_hid", Mv, self.info_branches[0][0].fc1.n)
        qb, kvb = tb("xc_q", Mv, Di), tb("xc_kv", Mv, 2 * Di)
        yn = [tb(f"xc_yn{v}", Mv, Di) for v in range(2)]
        rope = None
        if self.rope_freq:
            cos, sin = self._rope_tables(gh, gw)
            rope = (cos, sin, Np)
        xs = [y[:Mv], y[Mv:]]
        cols = lambda t, c0, c1: (t[0][:, c0:c1] if x3 else t[:, c0:c1])  # noqa: E731  (column view: hi plane of a split buffer)
        for layer in range(len(self.info_branches[0])):
            blks = [self.info_branches[v][layer] for v in range(2)]
            for v in range(2):  # the memory of view v's cross-attention: norm_y of the OTHER view's previous-layer tokens
                hip.layernorm(xs[1 - v], Di, None, Mv, Di, blks[v].nyw, blks[v].nyb, 1e-6, yn[v], split=x3)
            for v in range(2):
                w, x = blks[v], xs[v]
                # self-attention
                hip.layernorm(x, Di, None, Mv, Di, w.n1w, w.n1b, 1e-6, xn, split=x3)
                self.linear_rope(xn, w.qkv, Mv, qkv, rope, 2 * Di, gamma=w.qscale)
                if x3:
                    hip.attention_x3(qkv, ao, B, Np, heads, 0.125)
                else:
                    hip.attention(qkv, ao, B, Np, heads, 0.0 if w.qscale is not None else 0.125)
                self.linear(ao, w.proj, Mv, x, gamma=w.ls1, res=x)
                # cross-attention: queries from this view, keys / values from the other view's normed tokens
                hip.layernorm(x, Di, None, Mv, Di, w.nxw, w.nxb, 1e-6, xn, split=x3)
                self.linear_rope(xn, w.q, Mv, qb, rope, Di, gamma=w.qxscale)
                self.linear_rope(yn[v], w.kv, Mv, kvb, rope, Di)
                hip.cross_attention(cols(qb, 0, Di), Di, cols(kvb, 0, Di), cols(kvb, Di, 2 * Di), 2 * Di, cols(ao, 0, Di), Di, B, Np, Np, heads,
                                    0.0 if w.qxscale is not None else 0.125, fmt)
                self.linear(ao, w.projx, Mv, x, gamma=w.lsx, res=x)
                # MLP
                hip.layernorm(x, Di, None, Mv, Di, w.n2w, w.n2b, 1e-6, xn, split=x3)
                self.linear(xn, w.fc1, Mv, hid, act=hip.ACT_GELU)
                self.linear(hid, w.fc2, Mv, x, gamma=w.ls2, res=x)
            on_block(layer, y)

    def _encode(self, patches, B2: int, H: int, W: int):
        gh, gw = H // self.P, W // self.P
        Np, D = gh * gw, self.D
        N = Np + 1
        cls0, pos = self._pos_tables(H, W)
        x = self.buf("enc_x", (B2 * N, D))
        hip.fill_rows(x, D, B2, N, cls0, D)
        if self.numerics == "fast":
            hip.gemm_bf16(patches, self.pe_w, B2 * Np, D, KPAD, x, bias=self.pe_b, res=pos, res_row_mod=Np, out_row_group=Np)
        else:
            tmp = self.buf("pe_tmp", (B2 * Np, D))
            hip.conv2d(patches, 1, 1, B2 * Np, KPAD, self.pe_w, D, 1, 1, 1, 0, tmp, self.zero, bias=self.pe_b)
            hip.add_rows(tmp, D, pos, Np, x, D, Np, B2 * Np, D)
        return x, Np, N

    # ------------------------------------------------------------------ DPT head
    def _head(self, hw, tag: str, levels: List[torch.Tensor], level_dims: List[int], B: int, gh: int, gw: int, H: int, W: int, fork_from: Optional[torch.cuda.Stream] = None):
        """[U] DPTFeature + DPTRegressionProcessor + adaptors on NHWC maps.  ``hw``: one packed head (_Head) -> that head's
        adaptor outputs; or a _HeadG (several heads of identical layer shapes) -> every layer up to p_conv1 is ONE grouped launch
        on a head-major stacked batch of G * B images, then each head's own tail on its slice; returns a list of adaptor outputs
        in the order of ``hw.heads``.  Bit-identical either way (tests)."""
        G = getattr(hw, "groups", 1)
        Bt = G * B  # images in every activation buffer of this call
        Fd = hw.feature_dim
        ld = hw.layer_dims
        sizes = [(4 * gh, 4 * gw), (2 * gh, 2 * gw), (gh, gw), ((gh - 1) // 2 + 1, (gw - 1) // 2 + 1)]
        def chain(i: int):
            """Level i up to its layer_rn output: 1x1 projection, resize (ConvTranspose x4 / x2, identity, 3x3 stride 2), bias-free 3x3."""
            wsn = f"{tag}_l{i}"  # (a split-K workspace per concurrently running chain)
            lvl = levels[hw.hooks[i]]
            t = self.hbuf(f"{tag}_act{i}", (Bt, gh, gw, ld[i]))
            self.conv(lvl, B, gh, gw, hw.act[i][0], t, in_shared=G > 1, ws=wsn)  # every head reads the same pyramid level
            u = t
            if i < 2 or i == 3:
                u = self.hbuf(f"{tag}_post{i}", (Bt, sizes[i][0], sizes[i][1], ld[i]))
                self.conv(t, B, gh, gw, hw.act[i][1], u, ws=wsn)
            ri = self.hbuf(f"{tag}_rn{i}", (Bt, sizes[i][0], sizes[i][1], Fd))
            # split mode: the producer of an RCU input also writes relu(x) (ufm_conv2d_nhwc_bf16x3 out_relu), which takes
            # the ReLU out of the consumer's MFMA loop; the fp32 kernels apply it on their fragments (relu_in)
            rr = self.hbuf(f"{tag}_rn{i}_relu", (Bt, sizes[i][0], sizes[i][1], Fd)) if self.head_split else None
            self.conv(u, B, sizes[i][0], sizes[i][1], hw.rn[i], ri, out_relu=rr, ws=wsn)
            return (ri, rr)

        # The four level chains are independent until the fusion blocks meet them (coarse to fine).  At one or two pairs their
        # layers are latency-bound grids of a few dozen tiles: levels 2, 1, 0 run on three side streams (graph branches under
        # hipGraph capture) while this stream does level 3 and starts the fusion; every side stream is joined back right before
        # its level is read.  Bit-identical (same kernels on the same buffers).
        cur = torch.cuda.current_stream(self.dev)
        use_ls = self.level_streams and Bt <= self.level_streams_max_images and not hip.timer_serialises()
        if use_ls and fork_from is None and torch.cuda.is_current_stream_capturing():
            # Under hipGraph capture a fork from an ALREADY FORKED stream segfaults inside hipStreamEndCapture (ROCm 7.0 / 7.2 runtime,
            # profiles/r04/capture_nested_fork.log).  If this call runs on one of the engine's own forked streams (a head stream, a
            # micro-batch stream) and the caller did not say where that stream was forked from, the level chains stay serial
            # (same kernels, same buffers, same bits) instead of taking the process down.
            forked = [st for sl in self._head_streams.values() for st in sl] + list(self._streams)
            if any(cur == st for st in forked):
                use_ls = False
        r: List[Any] = [None] * 4
        lst: List[torch.cuda.Stream] = []
        if use_ls:
            lst = self._level_streams.setdefault((getattr(self._tls, "ns", ""), tag), [])
            while len(lst) < 3:
                lst.append(torch.cuda.Stream(device=self.dev))
            # fork_from: the stream this head's own stream was forked from (two-stream heads).  The level chains depend on the pyramid
            # only, so they fork from THAT stream: under hipGraph capture every fork then starts at the capture's origin stream --
            # a fork from an already forked stream (a nested fork) makes hipStreamEndCapture segfault on ROCm 7 (round-3 crash,
            # reproduced and bisected in round 4: tools/lab/capture_debug.py, profiles/r04/capture_nested_fork.log)
            src_stream = fork_from if fork_from is not None else cur
            for k, i in enumerate((2, 1, 0)):
                lst[k].wait_stream(src_stream)
                with torch.cuda.stream(lst[k]):
                    r[i] = chain(i)
            r[3] = chain(3)
        else:
            for i in range(4):
                r[i] = chain(i)

        def rcu(xs, pair, h, w, name, extra_res=None, want_relu=False):
            """[U] ResidualConvUnit: conv2(relu(conv1(relu(x)))) + x (+ extra_res).  xs = (x, relu(x) or None)."""
            x, xr = xs
            t1 = self.hbuf(f"{tag}_{name}_t", (Bt, h, w, Fd))
            o = self.hbuf(f"{tag}_{name}_o", (Bt, h, w, Fd))
            orl = self.hbuf(f"{tag}_{name}_or", (Bt, h, w, Fd)) if (want_relu and self.head_split) else None
            if self.head_split:
                self.conv(xr, B, h, w, pair[0], t1, act=hip.ACT_RELU, ws=tag)  # relu applied once, by the producers
                self.conv(t1, B, h, w, pair[1], o, res1=x, res2=extra_res, out_relu=orl, ws=tag)
            else:
                self.conv(x, B, h, w, pair[0], t1, relu_in=True, ws=tag)
                self.conv(t1, B, h, w, pair[1], o, relu_in=True, res1=x, res2=extra_res, ws=tag)
            return (o, orl)

        path = None
        for lvl in (3, 2, 1, 0):
            h, w = sizes[lvl]
            f = hw.fuse[lvl]
            if use_ls and lvl < 3:
                cur.wait_stream(lst[2 - lvl])  # level `lvl` ran on side stream 2 - lvl
            if path is None:
                s = r[lvl]
            else:
                s = rcu(r[lvl], f["r1"], h, w, f"f{lvl}a", extra_res=path, want_relu=True)  # path + resConfUnit1(r)
            o, _ = rcu(s, f["r2"], h, w, f"f{lvl}b")
            # out_conv (1x1) commutes with the bilinear x2 (weights sum to 1): run it at low resolution
            c = self.hbuf(f"{tag}_f{lvl}c", (Bt, h, w, Fd))
            self.conv(o, B, h, w, f["out"], c, ws=tag)
            if lvl == 3:
                th, tw = sizes[2]  # refinenet4 output is cropped to layer-3's grid
                path = self.hbuf(f"{tag}_p{lvl}", (Bt, th, tw, Fd))
                hip.upsample_bilinear(c, Bt, h, w, Fd, path, 2 * h, 2 * w, th, tw)
            else:
                path = self.hbuf(f"{tag}_p{lvl}", (Bt, 2 * h, 2 * w, Fd))
                hip.upsample_bilinear(c, Bt, h, w, Fd, path, 2 * h, 2 * w)
        h8, w8 = 2 * sizes[0][0], 2 * sizes[0][1]
        c1 = self.hbuf(f"{tag}_pc1", (Bt, h8, w8, hw.p_conv1.cout))
        self.conv(path, B, h8, w8, hw.p_conv1, c1, ws=tag)
        if G > 1:  # each head's own tail on its B images of the stacked p_conv1 output (callers checked _tail_fusable for every head)
            plane = Bt * h8 * w8 * hw.p_conv1.cout
            results = []
            for g, hh in enumerate(hw.heads):
                out = torch.empty((B, hh.tail_cout, H, W), device=self.dev, dtype=torch.float32)
                logits = torch.empty_like(out) if 1 in hh.kinds else None
                hip.dpt_tail_fused(c1[0][g * B : (g + 1) * B], B, h8, w8, 128, hh.p_conv2a.w, hh.p_conv2a.b, 32, H, W, hh.tail_w, hh.tail_b, hh.tail_cout, hh.kinds, hh.scale, hh.shift,
                                   out, logits, in_plane=plane)
                results.append(self._adaptor_outputs(hh, out, logits, B, H, W))
            return results
        out = torch.empty((B, hw.tail_cout, H, W), device=self.dev, dtype=torch.float32)
        logits = torch.empty_like(out) if 1 in hw.kinds else None
        c2a = hw.p_conv2a
        if self._tail_fusable(hw, gh, gw, H, W):
            # upsample -> conv3x3 + ReLU -> conv1x1 -> adaptor in one kernel: the two full-resolution maps stay on chip
            hip.dpt_tail_fused(c1, B, h8, w8, 128, c2a.w, c2a.b, 32, H, W, hw.tail_w, hw.tail_b, hw.tail_cout, hw.kinds, hw.scale, hw.shift, out, logits)
        else:
            up = self.hbuf(f"{tag}_up", (B, H, W, hw.p_conv1.cout))
            hip.upsample_bilinear(c1, B, h8, w8, hw.p_conv1.cout, up, H, W)
            c2 = self.hbuf(f"{tag}_pc2", (B, H, W, c2a.cout))
            self.conv(up, B, H, W, c2a, c2, act=hip.ACT_RELU, ws=tag)
            hip.head_tail(c2, B * H * W, H * W, hw.tail_cin, hw.tail_w, hw.tail_b, hw.tail_cout, hw.kinds, hw.scale, hw.shift, out, logits)
        return self._adaptor_outputs(hw, out, logits, B, H, W)

    def _tail_fusable(self, hw: _Head, gh: int, gw: int, H: int, W: int) -> bool:
        """ufm_dpt_tail_fused applies: split-format heads, the UFM-Base tail shape, an up-sampling ratio inside its halo-row budget."""
        h8, w8 = 8 * gh, 8 * gw
        c2a = hw.p_conv2a
        up_ok = h8 <= H and w8 <= W and H > 1 and W > 1 and max((h8 - 1) / (H - 1), (w8 - 1) / (W - 1)) * 17 + 2 <= 13
        return bool(self.fused_tail and self.head_split and up_ok and (c2a.cin, c2a.cout, c2a.k, c2a.stride, c2a.pad, hw.tail_cin) == (128, 32, 3, 1, 1, 32))

    def _adaptor_outputs(self, hw, out, logits, B: int, H: int, W: int):
        """Split the decoded channels between the adaptors ([U] AdaptorMap; ufm.py:644-660)."""
        res, c0 = {}, 0
        for a in hw.adaptors:
            n = a.required_channels
            raw = out[:, c0 : c0 + n]
            if a.cls_name == "Covariance2DAdaptor":  # ufm.py:648-651
                raw = raw.contiguous()
                cov, inv = torch.empty_like(raw), torch.empty_like(raw)
                logdet = torch.empty((B, 1, H, W), device=self.dev, dtype=torch.float32)
                hip.adaptor_covariance2d(raw, B, H * W, cov, inv, logdet)
                res[a.name] = dict(covariance=cov, inv_covariance=inv, log_det=logdet, kind=a.cls_name)
            elif a.cls_name == "FlowWithConfidenceAdaptor":  # ufm.py:38; only ``.value`` is read by the reference's forward
                craw = raw[:, 2:3].contiguous()
                conf = torch.empty_like(craw)
                hip.adaptor_confidence(craw, a.confidence_type, a.vmin, a.vmax, conf)
                res[a.name] = dict(value=raw[:, :2], confidence=conf, logits=None, kind=a.cls_name)
            elif a.cls_name == "ConfidenceAdaptor":  # ufm.py:653-654
                raw = raw.contiguous()
                val = torch.empty_like(raw)
                hip.adaptor_confidence(raw, a.confidence_type, a.vmin, a.vmax, val)
                res[a.name] = dict(value=val, logits=None, kind=a.cls_name)
            else:
                res[a.name] = dict(value=raw, logits=logits[:, c0 : c0 + n] if logits is not None else None, kind=a.cls_name)
            c0 += n
        return res

    # ------------------------------------------------------------------ MoGe convolutional head (head_type "moge_conv")
    def _head_moge(self, hw: _MoGeHead, tag: str, levels: List[torch.Tensor], B: int, gh: int, gw: int, H: int, W: int):
        """[U] MoGeConvFeature.forward on NHWC maps (oracle/uniception_ref.py::MoGeConvFeature; parity unpinned): per-level
        1x1 projections summed through the conv kernels' residual input; per stage {copy into a 32-channel-padded buffer +
        the two view-plane uv channels, ConvTranspose2d(k=s=2) as a pixel-shuffle GEMM, conv3x3 replicate, residual blocks
        (GroupNorm+ReLU kernel, conv, GroupNorm+ReLU, conv + skip)}; bilinear (align_corners=False) to (H, W) straight into
        the next concat buffer; conv3x3 replicate + ReLU; the last 1x1 conv + adaptor in ufm_head_tail."""
        aspect = W / H
        x = self.hbuf(f"{tag}_proj", (B, gh, gw, hw.dim_proj))
        for i, c in enumerate(hw.projects):  # torch.stack([...]).sum(dim=1): accumulated in level order
            self.conv(levels[i], B, gh, gw, c, x, res1=x if i > 0 else None)
        h, w = gh, gw

        def gn(t, spec, hh, ww, C, name):
            groups, gw_, gb_, eps = spec
            o = self.hbuf(name, (B, hh, ww, C))
            # sized for THIS call's map and group count (buf() re-allocates when a later call needs more): any number of
            # upsample stages / groups is covered, the kernel cannot check the workspace itself
            ws = self.buf(f"{tag}_gn_ws_{hh}x{ww}_{groups}", (hip.group_norm_ws_floats(B, hh * ww, groups),))
            hip.group_norm(t, B, hh * ww, C, groups, gw_, gb_, eps, True, o, ws)
            return o

        for k, st in enumerate(hw.stages):
            cat = self.hbuf(f"{tag}_cat{k}", (B, h, w, st["ldc"]))
            hip.resize_nearest(x, B, h, w, st["cin"], cat, h, w, st["ldc"], 0)   # torch.cat([x, uv], dim=1): the x slot
            hip.fill_uv(cat, B, h, w, st["ldc"], st["cin"], aspect)                 # uv + zero padding to the K chunk
            co = st["c3"].cout
            up = self.hbuf(f"{tag}_up{k}", (B, 2 * h, 2 * w, co))
            self.conv(cat, B, h, w, st["ct"], up)
            h, w = 2 * h, 2 * w
            x = self.hbuf(f"{tag}_c3_{k}", (B, h, w, co))
            self.conv(up, B, h, w, st["c3"], x, replicate=True)
            for j, rb in enumerate(st["res"]):
                t = gn(x, rb["gn1"], h, w, rb["a"].cin, f"{tag}_s{k}r{j}_n1")
                t2 = self.hbuf(f"{tag}_s{k}r{j}_a", (B, h, w, rb["a"].cout))
                self.conv(t, B, h, w, rb["a"], t2, replicate=True)
                t3 = gn(t2, rb["gn2"], h, w, rb["b"].cin, f"{tag}_s{k}r{j}_n2")
                skip = x
                if rb["skip"] is not None:
                    skip = self.hbuf(f"{tag}_s{k}r{j}_skip", (B, h, w, rb["b"].cout))
                    self.conv(x, B, h, w, rb["skip"], skip)
                o = self.hbuf(f"{tag}_s{k}r{j}_o", (B, h, w, rb["b"].cout))
                self.conv(t3, B, h, w, rb["b"], o, replicate=True, res1=skip)
                x = o
        cat = self.hbuf(f"{tag}_catf", (B, H, W, hw.out_ldc))
        hip.resize_bilinear(x, B, h, w, hw.out_cin, hw.out_cin, cat, H, W, hw.out_ldc, 0)
        hip.fill_uv(cat, B, H, W, hw.out_ldc, hw.out_cin, aspect)
        c = self.hbuf(f"{tag}_outc", (B, H, W, hw.out_conv.cout))
        self.conv(cat, B, H, W, hw.out_conv, c, act=hip.ACT_RELU, replicate=True)
        out = torch.empty((B, hw.tail_cout, H, W), device=self.dev, dtype=torch.float32)
        logits = torch.empty_like(out) if 1 in hw.kinds else None
        hip.head_tail(c, B * H * W, H * W, hw.tail_cin, hw.tail_w, hw.tail_b, hw.tail_cout, hw.kinds, hw.scale, hw.shift, out, logits)
        return self._adaptor_outputs(hw, out, logits, B, H, W)

    # ------------------------------------------------------------------ UNet fine features (UFM-Refine option)
    def _unet(self, img: torch.Tensor, N: int, H: int, W: int) -> torch.Tensor:
        """unet_encoder.py:50-71 on NHWC maps: (conv3x3+ReLU) x 2 / max-pool down, ConvTranspose(k=s=2) up, nearest fix-up
        of odd sizes, concat (skip | up), (conv3x3+ReLU) x 2, final 1x1.  ``img``: (N, H, W, 32) head-format buffer holding
        the normalised image in channels 0..2.  Returns (N, H, W, 32) with the 16 feature channels first."""
        un = self.unet
        x, h, w = img, H, W
        skips = []
        for i, (c1, c2) in enumerate(un["downs"]):  # :53-57
            t = self.hbuf(f"un_d{i}a", (N, h, w, c1.cout))
            self.conv(x, N, h, w, c1, t, act=hip.ACT_RELU)
            s = self.hbuf(f"un_d{i}b", (N, h, w, c2.cout))
            self.conv(t, N, h, w, c2, s, act=hip.ACT_RELU)
            skips.append((s, h, w, c2.cout))
            x = self.hbuf(f"un_p{i}", (N, h // 2, w // 2, c2.cout))
            hip.maxpool2x2(s, N, h, w, c2.cout, x)
            h, w = h // 2, w // 2
        c1, c2 = un["bottleneck"]
        t = self.hbuf("un_ba", (N, h, w, c1.cout))
        self.conv(x, N, h, w, c1, t, act=hip.ACT_RELU)
        x = self.hbuf("un_bb", (N, h, w, c2.cout))
        self.conv(t, N, h, w, c2, x, act=hip.ACT_RELU)
        for k, (ct, (c1, c2)) in enumerate(un["ups"]):  # :62-69
            skip, sh, sw, sc = skips[-1 - k]
            f = ct.cout // (ct.shuffle * ct.shuffle)
            up = self.hbuf(f"un_u{k}", (N, 2 * h, 2 * w, f))
            self.conv(x, N, h, w, ct, up)
            cat = self.hbuf(f"un_c{k}", (N, sh, sw, sc + f))
            hip.resize_nearest(skip, N, sh, sw, sc, cat, sh, sw, sc + f, 0)        # torch.cat((skip, x), dim=1): skip half
            hip.resize_nearest(up, N, 2 * h, 2 * w, f, cat, sh, sw, sc + f, sc)    # F.interpolate(x, size=skip.shape[2:]) | copy
            h, w = sh, sw
            t = self.hbuf(f"un_u{k}a", (N, h, w, c1.cout))
            self.conv(cat, N, h, w, c1, t, act=hip.ACT_RELU)
            x = self.hbuf(f"un_u{k}b", (N, h, w, c2.cout))
            self.conv(t, N, h, w, c2, x, act=hip.ACT_RELU)
        out = self.hbuf("un_out", (N, H, W, un["final"].cout))
        self.conv(x, N, H, W, un["final"], out)
        return out

    # ------------------------------------------------------------------ full forward
    @torch.no_grad()
    def forward(self, src, tgt, *, layout: int, scale3, shift3, H: int, W: int, Hs: int, Ws: int, Ht: int, Wt: int, symmetrized: bool = False) -> Dict[str, Any]:
        """src/tgt: device images (uint8 or float32, BHWC layout=0 / BCHW layout=1) of sizes
        (Hs,Ws)/(Ht,Wt); (H,W) is the network resolution.  Returns network-resolution outputs.

        Pairs are independent, so with ``micro_batches`` > 1 the batch is split into contiguous
        micro-batches that run CONCURRENTLY on separate HIP streams (one host thread each, own
        workspace): one micro-batch's HBM-bound phases and wave-quantization tails overlap the other's
        MFMA phases.  Results are identical to the single-stream run (each pair's arithmetic does not
        depend on its batch neighbours)."""
        self._pack()
        B = src.shape[0]
        if symmetrized and (B % 2 != 0 or (Hs, Ws) != (Ht, Wt)):
            raise ValueError("symmetrized=True needs an even number of equally sized pairs: (a,b),(b,a),... (ufm.py:336-352)")
        nmb = self.micro_batches if (B >= 2 * self.micro_batches and not hip.timer_serialises() and not symmetrized) else 1
        if nmb == 1:
            self._tls.ns = ""
            return self._forward_images(src, tgt, layout, scale3, shift3, H, W, Hs, Ws, Ht, Wt, symmetrized)
        bounds = [(i * B) // nmb for i in range(nmb + 1)]
        if self.mb_bounds is not None and self.mb_bounds[-1] == B:  # lab: explicit micro-batch boundaries (tools/lab/mb_split.py)
            bounds, nmb = list(self.mb_bounds), len(self.mb_bounds) - 1
        gh, gw = H // self.P, W // self.P
        self._pos_tables(H, W)  # shared read-only tables are built here, before the workers start
        if self.info_cross:
            if self.rope_freq:
                self._rope_tables(gh, gw)
        else:
            self._view_pe_table(gh * gw)
        for i in range(nmb):
            self._index_tables(bounds[i + 1] - bounds[i], gh * gw)
            if self.info_cross:
                self._view_major_tables(bounds[i + 1] - bounds[i], gh * gw)
        while len(self._streams) < nmb:
            self._streams.append(torch.cuda.Stream(device=self.dev))
            # micro-batch streams run side by side: tile heights for CU time, not latency (include/ufm_hip.h)
            if hip.hint_concurrent_stream(self._streams[-1]):
                self._flagged.append(self._streams[-1].cuda_stream)
            else:
                _warn_hint_refused()
        cur = torch.cuda.current_stream(self.dev)
        results: List[Any] = [None] * nmb
        errors: List[BaseException] = []

        # (UFM-Refine keeps per-micro-batch heads with UNet fine features -- its UNet input maps are built per micro-batch --
        #  and with the cross-attention variant, whose residual stream is view-major: a micro-batch is not a row slice of it)
        joint = self.joint_heads and not (self.refine and (self.unet is not None or self.info_cross))
        Np_ = gh * gw
        shared: List[torch.Tensor] = []
        y_full = enc_first_full = None
        if joint:  # full-batch pyramid buffers (main namespace); micro-batch i fills rows [bounds[i] * Np, bounds[i+1] * Np)
            self._tls.ns = ""
            shared = [self.hbuf(f"lvl_joint{k}", (B * Np_, d)) for k, d in enumerate([self.D, self.Di, self.Di, self.Di])]
            by_name = dict(zip(("lvl0", "lvl_i0", "lvl_i1", "lvl3"), shared))
            if self.refine:  # the classification head also reads the info-sharing residual stream and the first encoder level
                self._index_tables(B, Np_)  # the full-batch row tables, built before the workers start
                y_full = self.buf("info_x_joint", (B * 2 * Np_, self.Di))
                enc_first_full = (self.buf("cls_in_x2", (2, 2 * B * Np_, self.D + self.Di), torch.bfloat16) if self.head_split
                                  else self.buf("enc_first", (2 * B * Np_, self.D)))

        def work(i: int):
            try:
                with torch.cuda.device(self.dev), torch.cuda.stream(self._streams[i]):
                    self._tls.ns = f"mb{i}/"
                    lo, hi = bounds[i], bounds[i + 1]
                    # level_ln (and, for UFM-Refine, the residual stream / first encoder level) write their rows in place
                    self._tls.joint = dict(levels=by_name, row0=lo * Np_, rows_total=B * Np_, y=y_full, enc_first=enc_first_full,
                                           pair0=lo, pairs_total=B) if joint else None
                    try:
                        r = self._forward_images(src[lo:hi], tgt[lo:hi], layout, scale3, shift3, H, W, Hs, Ws, Ht, Wt, levels_only=joint)
                    finally:
                        self._tls.joint = None
                    results[i] = None if joint else r
            except BaseException as exc:  # surfaced on the caller's thread
                errors.append(exc)

        for s in self._streams[:nmb]:
            s.wait_stream(cur)  # inputs produced on the caller's stream
        threads = [threading.Thread(target=work, args=(i,)) for i in range(1, nmb)]
        for t in threads:
            t.start()
        work(0)
        for t in threads:
            t.join()
        for s in self._streams[:nmb]:
            cur.wait_stream(s)
        if errors:
            raise errors[0]
        if joint:
            self._tls.ns = ""
            tabs = None
            if self.refine:
                ix = self._index_tables(B, Np_)
                tabs = (ix["info_v1"], ix["info_v2"], ix["info_all"])
            return self._heads_and_refine(shared, [self.D, self.Di, self.Di, self.Di], B, H, W, y_full, enc_first_full, tabs)
        return self._merge(results)

    def _merge(self, parts: List[Dict[str, Any]]) -> Dict[str, Any]:
        """Concatenate micro-batch results along the pair dimension (small planar outputs only)."""
        def cat(vals):
            if vals[0] is None:
                return None
            if isinstance(vals[0], torch.Tensor):
                cur = torch.cuda.current_stream(self.dev)
                for v in vals:
                    v.record_stream(cur)  # allocated on a side stream, consumed here on the caller's stream
                return torch.cat(vals, dim=0)
            return vals[0]

        out: Dict[str, Any] = {}
        for tag in parts[0]:
            if tag == "refine":
                r = {k: cat([p[tag][k] for p in parts]) for k in parts[0][tag] if k != "feats"}
                f = [p[tag]["feats"] for p in parts]  # each (2b, C, H, W): [view-1 block | view-2 block]
                for x in f:
                    x.record_stream(torch.cuda.current_stream(self.dev))
                r["feats"] = torch.cat([x[: x.shape[0] // 2] for x in f] + [x[x.shape[0] // 2 :] for x in f], dim=0)
                out[tag] = r
            else:
                out[tag] = {name: {k: cat([p[tag][name][k] for p in parts]) for k in parts[0][tag][name]} for name in parts[0][tag]}
        return out

    def _forward_images(self, src, tgt, layout, scale3, shift3, H, W, Hs, Ws, Ht, Wt, symmetrized: bool = False, levels_only: bool = False):
        B = src.shape[0]
        gh, gw = H // self.P, W // self.P
        Np = gh * gw
        # symmetrized (ufm.py:345-347): only img1[::2] / img2[::2] are encoded; everything after the encoder sees all B pairs
        enc_views = ((src[::2].contiguous(), Hs, Ws), (tgt[::2].contiguous(), Ht, Wt)) if symmetrized else ((src, Hs, Ws), (tgt, Ht, Wt))
        Be = enc_views[0][0].shape[0]
        if symmetrized and (Hs, Ws) != (H, W):
            raise NotImplementedError("symmetrized=True is the lower-level forward() path: images arrive at network resolution")
        patches = self.buf("patches", (2 * Be * Np, KPAD), self.adt)
        want_unet = self.refine and self.unet is not None
        # view["img"] of ufm.py:915-917 per view: normalised, network resolution, NHWC with 3 -> 32 zero-padded channels
        unet_imgs = [self.hbuf(f"un_img{v}", (B, H, W, 32)) for v in range(2)] if want_unet else None
        for v, (img, h0, w0) in enumerate(enc_views):
            dst = patches[v * Be * Np : (v + 1) * Be * Np]
            full = (src, tgt)[v]
            if (h0, w0) == (H, W):
                hip.patchify(img, layout, Be, H, W, self.P, scale3, shift3, dst, KPAD)
                if want_unet:
                    hip.image_to_nhwc(full, layout, B, H, W, scale3, shift3, unet_imgs[v], 32)
            else:  # normalise-on-load + separable antialias resize (flow_resizing.py:313-326), then patchify
                rs = self.buf(f"resized{v}", (B, 3, H, W))
                tmp = self.buf(f"resize_tmp{v}", (B * 3 * h0 * W,))
                hip.resize_antialias(img, layout, B, h0, w0, scale3, shift3, rs, H, W, tmp)
                hip.patchify(rs, 1, B, H, W, self.P, [1.0, 1.0, 1.0], [0.0, 0.0, 0.0], dst, KPAD)
                if want_unet:
                    hip.image_to_nhwc(rs, 1, B, H, W, [1.0, 1.0, 1.0], [0.0, 0.0, 0.0], unet_imgs[v], 32)
        return self._forward_patches(patches, B, H, W, unet_imgs, symmetrized, levels_only)

    def _forward_patches(self, patches, B: int, H: int, W: int, unet_imgs: Optional[List[torch.Tensor]] = None, symmetrized: bool = False, levels_only: bool = False):
        B2 = 2 * B
        gh, gw = H // self.P, W // self.P
        n_enc = B if symmetrized else B2  # images that go through the encoder
        x, Np, N = self._encode(patches, n_enc, H, W)
        idx = self._index_tables(B, Np, symmetrized)
        D, Di = self.D, self.Di
        nw, nb = self.enc_norm
        enc_first = None
        last_i = len(self.enc_blocks) - 1
        lvl0 = None
        enc_info = self.buf("enc_info", (B * 2 * Np, D), self.adt)
        # info-sharing token order: global attention (pair, view, patch) -- one pair's joint tokens contiguous; the
        # cross-attention variant (view, pair, patch) -- one view's tokens of the whole batch contiguous
        enc_info_rows = idx["enc_all"] if self.info_cross else idx["enc_info"]

        jt = getattr(self._tls, "joint", None)

        def on_enc(i, xx):
            nonlocal enc_first, lvl0
            if self.refine and i == self.enc_indices[0] and jt is not None and jt["enc_first"] is not None:
                # joint heads: this micro-batch's view-1 / view-2 rows of the full-batch buffer ([view-1 batch | view-2 batch])
                full, Bt, p0 = jt["enc_first"], jt["pairs_total"], jt["pair0"]
                ld = full.shape[-1]
                for v, tab in enumerate((idx["enc_v1"], idx["enc_all"][B * Np :])):
                    dst = full.narrow(-2, (v * Bt + p0) * Np, B * Np)
                    if self.head_split:
                        hip.layernorm(xx, D, tab, B * Np, D, nw, nb, 1e-6, dst[0], ldo=ld, split=True, out_plane=2 * Bt * Np * ld)
                    else:
                        hip.layernorm(xx, D, tab, B * Np, D, nw, nb, 1e-6, dst)
                enc_first = full
            elif self.refine and i == self.enc_indices[0]:
                if self.head_split:  # straight into the classification head's split-format input, columns [0, D)
                    enc_first = self.buf("cls_in_x2", (2, B2 * Np, D + Di), torch.bfloat16)
                    hip.layernorm(xx, D, idx["enc_all"], B2 * Np, D, nw, nb, 1e-6, enc_first[0], ldo=D + Di, split=True)
                else:
                    enc_first = self.buf("enc_first", (B2 * Np, D))
                    hip.layernorm(xx, D, idx["enc_all"], B2 * Np, D, nw, nb, 1e-6, enc_first)
            if i == self.enc_indices[-1]:
                hip.layernorm(xx, D, enc_info_rows, B * 2 * Np, D, nw, nb, 1e-6, enc_info)
                lvl0 = self.level_ln(xx, D, idx["enc_v1"], B * Np, nw, nb, "lvl0")

        blocks = self.enc_blocks[: self.enc_indices[-1] + 1]  # blocks past the last returned index never matter
        self._blocks(blocks, x, n_enc, N, D, self.enc_heads, on_enc, needs_x=lambda i: (self.refine and i == self.enc_indices[0]) or i == self.enc_indices[-1])

        # ---- info sharing ----
        M2 = B * 2 * Np
        if jt is not None and jt["y"] is not None:  # joint heads (UFM-Refine): this micro-batch's rows of the full-batch stream
            y = jt["y"].narrow(0, jt["pair0"] * 2 * Np, M2)
        else:
            y = self.buf("info_x", (M2, Di))
        vpe = None if self.info_cross else self._view_pe_table(Np)
        if self.info_proj is None:
            src32 = enc_info if enc_info.dtype == torch.float32 else None
            if src32 is None:
                src32 = self.buf("enc_info32", (M2, D))
                hip.layernorm(x, D, enc_info_rows, M2, D, nw, nb, 1e-6, src32)
            hip.add_rows(src32, D, vpe, 2 * Np, y, Di, 0, M2, Di)
        elif self.numerics == "fast":
            hip.gemm_bf16(enc_info, self.info_proj.w, M2, Di, D, y, bias=self.info_proj.b, res=vpe, res_row_mod=2 * Np if vpe is not None else 0)
        elif vpe is None:
            self.linear(enc_info, self.info_proj, M2, y)
        else:
            tmp = self.buf("info_tmp", (M2, Di))
            self.linear(enc_info, self.info_proj, M2, tmp)
            hip.add_rows(tmp, Di, vpe, 2 * Np, y, Di, 0, M2, Di)
        inw, inb = self.info_norm
        inter: List[torch.Tensor] = []
        if self.info_cross:
            iv = self._view_major_tables(B, Np)
            info_v1, info_v2, info_all = iv["v1"], iv["v2"], iv["all"]
        else:
            info_v1, info_v2, info_all = idx["info_v1"], idx["info_v2"], idx["info_all"]

        def on_info(i, yy):
            if i in self.info_indices:
                inter.append(self.level_ln(yy, Di, info_v1, B * Np, inw, inb, f"lvl_i{len(inter)}"))

        nlast = len(self.info_blocks) - 1 if not self.info_cross else -1
        half_last = (self.last_layer_view1 and not self.info_cross and not self.refine and self.numerics == "fast" and not self.defer_residual
                     and nlast >= 1 and nlast not in self.info_indices and self.info_blocks[nlast].qscale is not None)
        y_last, last_rows = y, info_v1
        if self.info_cross:
            self._cross_blocks(y, B, Np, gh, gw, on_info)
        elif half_last:
            self._blocks(self.info_blocks[:nlast], y, B, 2 * Np, Di, self.info_heads, on_info, needs_x=lambda i: i in self.info_indices or i == nlast - 1)
            y_last, last_rows = self._last_block_view1(self.info_blocks[nlast], y, B, Np, Di, self.info_heads, info_v1), None
        else:
            self._blocks(self.info_blocks, y, B, 2 * Np, Di, self.info_heads, on_info, needs_x=lambda i: i in self.info_indices)
        if len(inter) != 2:
            raise ValueError("info_sharing.indices must name two blocks (ufm.py:605-606 reads intermediates [0] and [1])")
        lvl3 = self.level_ln(y_last, Di, last_rows, B * Np, inw, inb, "lvl3")
        levels = [lvl0, inter[0], inter[1], lvl3]  # ufm.py:603-608 (view-1 pyramid only; view 2's is never decoded)
        dims = [D, Di, Di, Di]
        if levels_only:  # joint_heads: the caller runs the heads on the whole batch
            return levels
        return self._heads_and_refine(levels, dims, B, H, W, y, enc_first, (info_v1, info_v2, info_all), unet_imgs)

    def _heads_and_refine(self, levels, dims, B: int, H: int, W: int, y=None, enc_first=None, info_tabs=None, unet_imgs=None) -> Dict[str, Any]:
        """The prediction heads on the view-1 pyramid (ufm.py:637-660) and, for UFM-Refine, the classification features +
        local refinement (ufm.py:949-1007; needs the trunk's residual stream ``y`` and ``enc_first``)."""
        B2 = 2 * B
        gh, gw = H // self.P, W // self.P
        Np = gh * gw
        D, Di = self.D, self.Di
        inw, inb = self.info_norm
        info_v1, info_v2, info_all = info_tabs if info_tabs is not None else (None, None, None)
        out: Dict[str, Any] = {}

        def run_head(hw, tag, fork_from=None):
            if isinstance(hw, _MoGeHead):
                return self._head_moge(hw, tag, levels, B, gh, gw, H, W)
            return self._head(hw, tag, levels, dims, B, gh, gw, H, W, fork_from=fork_from)

        hlist = list(self.heads.values())
        if (self.group_heads and len(hlist) > 1 and self.head_split and _HeadG.compatible(hlist) and hlist[0].p_conv1.cout == 128
                and all(self._tail_fusable(h, gh, gw, H, W) for h in hlist)):
            # the heads are the same graph with different weights (ufm.py:553-556, 637-642): one grouped launch per layer
            if self._head_group is None:
                self._head_group = _HeadG(hlist)
            for tag, res in zip(self.heads.keys(), self._head(self._head_group, "hg", levels, dims, B, gh, gw, H, W)):
                out[tag] = res
            conc = None
        else:
            conc = self.concurrent_heads if self.concurrent_heads is not None else getattr(self._tls, "ns", "") == ""
        if conc is None:
            pass
        elif len(self.heads) > 1 and conc and not hip.timer_serialises():
            # the heads only share their (read-only) input pyramid: run them on separate HIP streams so the
            # latency-bound small-grid layers of one overlap the large layers of the other
            main = torch.cuda.current_stream(self.dev)
            side_streams = self._head_streams.setdefault(getattr(self._tls, "ns", ""), [])
            while len(side_streams) < len(self.heads) - 1:
                side_streams.append(torch.cuda.Stream(device=self.dev))
            tags = list(self.heads.items())
            for (tag, hw), st in zip(tags[1:], side_streams):
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    out[tag] = run_head(hw, tag, fork_from=main)
            out[tags[0][0]] = run_head(tags[0][1], tags[0][0])
            for (tag, _), st in zip(tags[1:], side_streams):
                main.wait_stream(st)
                for v in out[tag].values():
                    for t in v.values():
                        if isinstance(t, torch.Tensor):
                            t.record_stream(main)
        else:
            for tag, hw in self.heads.items():
                out[tag] = run_head(hw, tag)

        if self.refine:  # ufm.py:949-1007
            C1 = D + Di
            if self.head_split:
                # torch.cat([enc_first | info_final], channels), views stacked on batch (ufm.py:955-964), built in the split
                # format by the two LayerNorm launches themselves; the MLP is two 1x1 bf16x3 convolutions over the rows
                cat = enc_first
                hip.layernorm(y, Di, info_all, B2 * Np, Di, inw, inb, 1e-6, cat[0][:, D:], ldo=C1, split=True)
                hidden = self.buf("cls_hid_x2", (2, B2 * Np, self.cls_fc1.n), torch.bfloat16)
                self.conv(cat, 1, 1, B2 * Np, self.cls_fc1, hidden, act=hip.ACT_GELU)
                tok = self.buf("cls_tok_x2", (2, B2 * Np, self.cls_fc2.n), torch.bfloat16)
                self.conv(hidden, 1, 1, B2 * Np, self.cls_fc2, tok)
            else:
                lvl3a = self.buf("lvl3_v1_f32", (B * Np, Di))
                hip.layernorm(y, Di, info_v1, B * Np, Di, inw, inb, 1e-6, lvl3a)
                lvl3b = self.buf("lvl3_v2", (B * Np, Di))
                hip.layernorm(y, Di, info_v2, B * Np, Di, inw, inb, 1e-6, lvl3b)
                cat = self.buf("cls_in", (B2 * Np, C1))
                # torch.cat along channels == strided row copies: enc_first | info_final, views stacked on batch
                hip.add_rows(enc_first, D, None, 0, cat, C1, 0, B2 * Np, D)
                hip.add_rows(lvl3a, Di, None, 0, cat[:, D:], C1, 0, B * Np, Di)
                hip.add_rows(lvl3b, Di, None, 0, cat[B * Np :, D:], C1, 0, B * Np, Di)
                hidden = self.buf("cls_hid", (B2 * Np, self.cls_fc1.n))
                self.linear(cat, self.cls_fc1, B2 * Np, hidden, act=hip.ACT_GELU)
                tok = self.buf("cls_tok", (B2 * Np, self.cls_fc2.n))
                self.linear(hidden, self.cls_fc2, B2 * Np, tok)
            feats = torch.empty((B2, self.cls_out_dim, H, W), device=self.dev)
            hip.pixel_shuffle_planar(tok, B2, gh, gw, self.cls_out_dim, self.P, feats, split=self.head_split)
            if self.unet is not None:  # ufm.py:915-917, :967-983: one UNet pass per view, then the per-pixel combine
                combined = torch.empty_like(feats)
                for v in range(2):
                    uf = self._unet(unet_imgs[v], B, H, W)
                    hip.unet_combine(feats[v * B : (v + 1) * B], uf, B, H * W, uf.shape[-1], self.unet_w1, self.unet_b1, self.unet_w2, self.unet_b2,
                                     self.unet_method, combined[v * B : (v + 1) * B])
                feats = combined
            flow = out["head1"]["flow"]["value"]
            residual = torch.empty_like(flow)
            m = self.model
            logp = torch.empty((B, H, W, m.refinement_range, m.refinement_range), device=self.dev)
            hip.refine(flow.contiguous(), feats, B, self.cls_out_dim, H, W, m.refinement_range, float(m.temperature), self.cls_bias, residual, logp)
            refined = torch.empty_like(flow)
            hip.add_f32(flow.contiguous(), residual, refined)
            out["refine"] = dict(flow=refined, residual=residual, log_softmax=logp, feats=feats)
        return out
