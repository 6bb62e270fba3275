"""Per-kernel parity tests: every C-ABI entry point of libufm_hip.so against a plain PyTorch fp32
CPU statement of the same op (and the oracle / reference goldens where they exist).
Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""

import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from ufm_amd import hip as h

    h.lib()  # fail loudly if the extension is missing
    return h


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def bf16r(x):
    return x.to(torch.bfloat16).float()


# ----------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize(
    "M,N,K,act,use_gamma,use_res,out_bf16,row_mod,row_group",
    [
        (300, 256, 128, 0, False, False, True, 0, 0),
        (300, 256, 128, 1, False, False, True, 0, 0),
        (257, 128, 64, 0, True, True, False, 0, 0),
        (2 * 1370, 3072, 1024, 0, False, False, True, 0, 0),
        (2 * 1370, 1024, 4096, 0, True, True, False, 0, 0),
        (4 * 16, 128, 640, 0, False, True, False, 16, 16),  # patch-embed style: pos-embed table + cls slot skip
    ],
)
def test_gemm_bf16(hip, M, N, K, act, use_gamma, use_res, out_bf16, row_mod, row_group):
    A = bf16r(rnd(M, K, seed=1))
    W = bf16r(rnd(N, K, seed=2, scale=K**-0.5))
    bias = rnd(N, seed=3, scale=0.1)
    gamma = 1 + rnd(N, seed=4, scale=0.1) if use_gamma else None
    rows_res = row_mod if row_mod else M
    res = rnd(rows_res, N, seed=5) if use_res else None
    ref = A.double() @ W.double().T + bias.double()
    if act == 1:
        ref = F.gelu(ref)
    if gamma is not None:
        ref = ref * gamma.double()
    if res is not None:
        ref = ref + (res.double()[torch.arange(M) % rows_res] if row_mod else res.double())
    out_rows = M + M // row_group if row_group else M
    out = torch.full((out_rows, N), 7.0, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=DEV)
    res_d = res.to(DEV) if res is not None else None
    hip.gemm_bf16(
        A.to(DEV).bfloat16(), W.to(DEV).bfloat16(), M, N, K, out, bias=bias.to(DEV), act=act,
        gamma=gamma.to(DEV) if gamma is not None else None, res=res_d, res_row_mod=row_mod, out_row_group=row_group,
    )
    got = out.float().cpu().double()
    if row_group:
        idx = torch.arange(M)
        orow = (idx // row_group) * (row_group + 1) + 1 + idx % row_group
        assert torch.all(got[0 :: row_group + 1] == 7.0), "cls slots must be untouched"
        got = got[orow]
    tol = 2e-2 if out_bf16 else 2e-4  # bf16 output rounding (|x|~1..4) vs fp32-accumulate error only
    assert (got - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(2740, 1024, 4096), (2738, 768, 768), (1370, 1024, 320), (700, 3072, 1024)])
def test_gemm_128_two_k_tiles_per_barrier_is_bit_identical(hip, M, N, K):
    """Grids of at most one block per CU (the one-pair shapes) run the 128x128 kernel with two K-tiles per barrier
    (gemm_bf16.hip KS = 2, 128 KiB of LDS): same MFMA sequence per accumulator, so it must equal the one-tile-per-barrier
    form (flag 128) bit for bit -- even K-tile counts, an odd one (K = 320: 5 tiles, the last group half full), all four
    epilogue forms' outputs (fp32 residual, bf16 GELU)."""
    lib = hip.lib()
    A = bf16r(rnd(M, K, seed=1)).to(DEV).bfloat16()
    W = bf16r(rnd(N, K, seed=2, scale=K**-0.5)).to(DEV).bfloat16()
    bias, res = rnd(N, seed=3, scale=0.1).to(DEV), rnd(M, N, seed=5).to(DEV)
    outs = {}
    try:
        lib.ufm_debug_set_gemm_variant(1)
        for flag in (128, 0):
            lib.ufm_debug_set_gemm_flags(flag)
            f = torch.full((M, N), 3.0, device=DEV)
            hip.gemm_bf16(A, W, M, N, K, f, bias=bias, res=res)
            g = torch.full((M, N), 3.0, device=DEV, dtype=torch.bfloat16)
            hip.gemm_bf16(A, W, M, N, K, g, bias=bias, act=1)
            outs[flag] = (f, g)
    finally:
        lib.ufm_debug_set_gemm_variant(0)
        lib.ufm_debug_set_gemm_flags(0)
    assert torch.equal(outs[0][0], outs[128][0])
    assert torch.equal(outs[0][1].view(torch.int16), outs[128][1].view(torch.int16))
    ref = A.float() @ W.float().T + bias + res
    assert (outs[0][0] - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(256 * 9 + 77, 1024, 128), (256 * 7, 1024, 192), (5000, 2304, 768), (4 * 1370, 1024, 4096), (300, 256, 1024), (8 * 1369, 768, 768),
                                   (256 * 5 + 3, 640, 256), (2000, 384, 320), (1370 * 3, 1024, 448), (777, 128, 832)])
def test_gemm_8phase_bitwise_equals_128_kernel_repeated(hip, M, N, K):
    """Race screen for the counted-vmcnt schedule: the 8-phase kernel (variant 4), the hybrid split (5) and the 256x128
    two-resident-workgroups kernel (6, round 5) accumulate in the same K order as the 128x128 kernel (1), so all must agree
    BIT FOR BIT -- over repeated launches (a DMA that lands late shows up as a rare wrong tile), K-tile counts 2 (prologue +
    drain only), 3 (odd), 4, 5, 7, 12, 13, 16, 64 (every exit of the pair kernel's six-body loop), ragged M, N % 256 != 0."""
    lib = hip.lib()
    A = bf16r(rnd(M, K, seed=1)).to(DEV).bfloat16()
    W = bf16r(rnd(N, K, seed=2, scale=K**-0.5)).to(DEV).bfloat16()
    bias, res = rnd(N, seed=3, scale=0.1).to(DEV), rnd(M, N, seed=5).to(DEV)
    try:
        lib.ufm_debug_set_gemm_variant(1)
        want = torch.zeros(M, N, device=DEV)
        hip.gemm_bf16(A, W, M, N, K, want, bias=bias, res=res)
        want_b = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
        hip.gemm_bf16(A, W, M, N, K, want_b, bias=bias, act=1)
        # the other two compile-time epilogue forms of the 8-phase kernel (gemm_common.h EpiTraits 2 and 3): bias + per-column
        # scale -> bf16 (QKV with the Q pre-scale), bias + LayerScale + fp32 residual (proj / fc2)
        gamma = (1.0 + rnd(N, seed=7, scale=0.2)).to(DEV)
        want_g = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
        hip.gemm_bf16(A, W, M, N, K, want_g, bias=bias, gamma=gamma)
        want_gr = torch.zeros(M, N, device=DEV)
        hip.gemm_bf16(A, W, M, N, K, want_gr, bias=bias, gamma=gamma, res=res)
        # (variant, pinned tile rows): the 8-phase kernel at its four tile heights (160 / 192 / 224 rows skip the MFMA
        # fragments past the tile's end; ragged last tiles at every height), the hybrid split and the cost model's own pick
        for rep in range(3):
            # (1, -128): the 128x128 kernel with the SERIAL read-modify-write read-out of rounds 1-4 (flag 0x800000) -- `want` above comes
            # from round 5's pipelined one (residual loads out of the store chain, gemm_common.h epilogue_lds_rmw2): same bits
            for variant, rows in ((4, 256), (4, 224), (4, 192), (4, 160), (5, 0), (0, 0), (6, 0), (6, -3), (6, 160), (6, 192), (6, 224), (1, -128), (4, -128), (6, -128)):
                if variant in (4, 5) and N % 256:
                    continue  # (6, rows): the pair kernel at the lower tile heights
                lib.ufm_debug_set_gemm_variant(variant)
                lib.ufm_debug_set_gemm_tile_rows(max(rows, 0))
                lib.ufm_debug_set_gemm_flags((-rows) << 16 if rows < 0 else 0)  # (6, -3): second resident workgroups start 3 sleeps late; -128 << 16 = 0x800000
                got = torch.full((M, N), 3.0, device=DEV)
                hip.gemm_bf16(A, W, M, N, K, got, bias=bias, res=res)
                assert torch.equal(got, want), (variant, rows, rep)
                got_b = torch.full((M, N), 3.0, device=DEV, dtype=torch.bfloat16)
                hip.gemm_bf16(A, W, M, N, K, got_b, bias=bias, act=1)
                assert torch.equal(got_b.view(torch.int16), want_b.view(torch.int16)), (variant, rows, rep)
                if rep == 0:
                    got_g = torch.full((M, N), 3.0, device=DEV, dtype=torch.bfloat16)
                    hip.gemm_bf16(A, W, M, N, K, got_g, bias=bias, gamma=gamma)
                    assert torch.equal(got_g.view(torch.int16), want_g.view(torch.int16)), (variant, rows, "bias+gamma -> bf16")
                    got_gr = torch.full((M, N), 3.0, device=DEV)
                    hip.gemm_bf16(A, W, M, N, K, got_gr, bias=bias, gamma=gamma, res=res)
                    assert torch.equal(got_gr, want_gr), (variant, rows, "bias+gamma+residual")
    finally:
        lib.ufm_debug_set_gemm_variant(0)
        lib.ufm_debug_set_gemm_tile_rows(0)
        lib.ufm_debug_set_gemm_flags(0)


@pytest.mark.parametrize("M,N,K,mode", [(10952, 2304, 768, "scale"), (21904, 2304, 768, "scale"), (10952, 3072, 768, "gelu"), (10960, 3072, 1024, "scale"),
                                         (21904, 768, 768, "rmw"), (10952, 768, 768, "rmw"), (21904, 768, 3072, "rmw"), (10952, 768, 3072, "rmw")])
def test_gemm_auto_dispatch_on_the_pipeline_shapes_bitwise_equals_128_kernel(hip, M, N, K, mode):
    """The auto dispatch at the shapes its per-shape rules name (the 256x128 two-resident-workgroups kernel on the D = 768 widths and on the
    encoder's QKV at micro-batch rows, round 5; the hybrid / persistent forms elsewhere): whatever it picks is BIT-identical to the 128x128
    kernel, with the rules on (default) and flipped (flag bits 24..27), for the three hot epilogues, ragged last tiles included."""
    lib = hip.lib()
    A = bf16r(rnd(M, K, seed=1)).to(DEV).bfloat16()
    W = bf16r(rnd(N, K, seed=2, scale=K**-0.5)).to(DEV).bfloat16()
    bias, gamma = rnd(N, seed=3, scale=0.1).to(DEV), (1.0 + rnd(N, seed=7, scale=0.2)).to(DEV)
    res0 = rnd(M, N, seed=9).to(DEV)

    def run():
        if mode == "rmw":
            out = res0.clone()
            hip.gemm_bf16(A, W, M, N, K, out, bias=bias, gamma=gamma, res=out)
        else:
            out = torch.full((M, N), 3.0, device=DEV, dtype=torch.bfloat16)
            hip.gemm_bf16(A, W, M, N, K, out, bias=bias, act=1 if mode == "gelu" else 0, gamma=None if mode == "gelu" else gamma)
        return out

    try:
        lib.ufm_debug_set_gemm_variant(1)
        want = run()
        lib.ufm_debug_set_gemm_variant(0)
        for flags in (0, 15 << 24, 8 << 24):
            lib.ufm_debug_set_gemm_flags(flags)
            for rep in range(2):
                got = run()
                assert torch.equal(got.view(torch.uint8), want.view(torch.uint8)), (flags, rep)
    finally:
        lib.ufm_debug_set_gemm_variant(0)
        lib.ufm_debug_set_gemm_flags(0)


def test_gemm_lab_splitk_is_deterministic_and_batch_invariant(hip):
    """Round 6 lab arm (ufm_debug_set_gemm_splitk): the read-modify-write launch as two K halves per 256 x 256 tile, the second
    workgroup to arrive adds the halves in half order.  The cut (K / 2) depends on the layer only: (i) repeated launches agree bit for bit whoever
    arrives last, (ii) the first rows of a large batch equal the same rows computed alone (batch invariance), (iii) against the unsplit kernel the
    result differs in the last bits only, (iv) the counters are left zero (a second launch on the same workspace works)."""
    import ctypes as C

    lib = hip.lib()
    N, K = 1024, 4096
    M_big, M_small = 10960, 2740
    A = bf16r(rnd(M_big, K, seed=1)).to(DEV).bfloat16()
    W = bf16r(rnd(N, K, seed=2, scale=K**-0.5)).to(DEV).bfloat16()
    bias, gamma = rnd(N, seed=3, scale=0.1).to(DEV), (1.0 + rnd(N, seed=7, scale=0.2)).to(DEV)
    res0 = rnd(M_big, N, seed=9).to(DEV)
    st = torch.cuda.current_stream()
    ws = torch.zeros((65536 + 200 * 2 * 262144) // 4, device=DEV)

    def run(M):
        o = res0[:M].clone()
        hip.gemm_bf16(A[:M], W, M, N, K, o, bias=bias, gamma=gamma, res=o)
        return o

    plain = run(M_big)
    try:
        assert lib.ufm_debug_set_gemm_splitk(C.c_void_p(st.cuda_stream), C.c_void_p(ws.data_ptr()), ws.numel() * 4, 3072) == 0
        a = run(M_big)
        for rep in range(4):
            assert torch.equal(run(M_big), a), rep
        small = run(M_small)
        assert torch.equal(small, a[:M_small])
        assert int(ws[:16384].view(torch.int32).abs().sum()) == 0
    finally:
        assert lib.ufm_debug_set_gemm_splitk(C.c_void_p(st.cuda_stream), None, 0, 256) == 0
    assert not torch.equal(a, plain) and (a - plain).abs().max().item() <= 2e-5 * plain.abs().max().item()
    assert torch.equal(run(M_big), plain)  # entry removed: the unsplit kernel again


def test_concurrent_stream_hint_changes_the_dispatch_not_the_bits(hip):
    """ufm_hint_concurrent_stream (round 5): on a flagged stream launches of 8192 rows or more use full-height 8-phase tiles only (CU time
    instead of latency as the objective).  Same arithmetic in the same order: the read-modify-write proj shape at micro-batch rows (172 tiles
    of 256 rows flagged, 232 of 192 rows unflagged) and the bf16x3 Linear form give the same bits on a flagged and an unflagged stream; the
    flag can be removed; the null stream is refused."""
    lib = hip.lib()
    M, N, K = 10960, 1024, 1024
    A = bf16r(rnd(M, K, seed=1)).to(DEV).bfloat16()
    W = bf16r(rnd(N, K, seed=2, scale=K**-0.5)).to(DEV).bfloat16()
    bias, gamma = rnd(N, seed=3, scale=0.1).to(DEV), (1.0 + rnd(N, seed=7, scale=0.2)).to(DEV)
    res0 = rnd(M, N, seed=9).to(DEV)
    Ax = torch.stack([A, torch.zeros_like(A)]).contiguous()
    Wx = torch.stack([W, torch.zeros_like(W)]).contiguous()
    zero = torch.zeros(256, device=DEV)
    side = torch.cuda.Stream()
    assert lib.ufm_hint_concurrent_stream(None, 1) != 0
    outs = {}
    try:
        for flagged in (False, True, False):
            assert hip.hint_concurrent_stream(side, flagged)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                o = res0.clone()
                hip.gemm_bf16(A, W, M, N, K, o, bias=bias, gamma=gamma, res=o)
                o3 = res0.clone()
                hip.gemm_x3(Ax, Wx, M, N, K, o3, zero, bias=bias, gamma=gamma, res=o3)
            side.synchronize()
            outs.setdefault(flagged, []).append((o, o3))
    finally:
        hip.hint_concurrent_stream(side, False)
    (a, a3), (c, c3) = outs[False]
    (b, b3), = outs[True]
    assert torch.equal(a.view(torch.int32), b.view(torch.int32)) and torch.equal(a.view(torch.int32), c.view(torch.int32))
    assert torch.equal(a3.view(torch.int32), b3.view(torch.int32)) and torch.equal(a3.view(torch.int32), c3.view(torch.int32))


@pytest.mark.parametrize("M,N,K", [(256 * 9, 1024, 128), (256 * 40, 2048, 192), (256 * 32, 4096, 1024), (256 * 86, 3072, 1024), (256 * 3, 256, 256), (256 * 65, 1024, 448)])
def test_gemm_persistent_8phase_bitwise_equals_128_kernel_repeated(hip, M, N, K):
    """gemm_bf16_8ph_persist.hip (round 5): one workgroup per CU walks tiles v, v + grid, ..., the next tile's prologue DMAs are issued in
    front of the current tile's epilogue stores and the epilogue goes through a 4-KiB staging slice per wave.  Same K order and the same
    operations per element as the other kernels: BIT-identical to the 128x128 kernel, for both bf16-output epilogues (bias + GELU, bias +
    per-column scale), over repeated launches (a prefetch that lands in a buffer still being read, or a counted wait that a store of the
    previous tile makes too weak, shows as a rare wrong tile), with 1 / 2 / 3+ tiles per workgroup, uneven tile counts (variant 7 runs the
    kernel on any tile count) and whole rounds (the shapes the auto dispatch sends to it), K-tile counts 2, 3, 7, 16."""
    lib = hip.lib()
    A = bf16r(rnd(M, K, seed=1)).to(DEV).bfloat16()
    W = bf16r(rnd(N, K, seed=2, scale=K**-0.5)).to(DEV).bfloat16()
    bias, gamma = rnd(N, seed=3, scale=0.1).to(DEV), (1.0 + rnd(N, seed=7, scale=0.2)).to(DEV)
    try:
        lib.ufm_debug_set_gemm_variant(1)
        want_b = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
        hip.gemm_bf16(A, W, M, N, K, want_b, bias=bias, act=1)
        want_g = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
        hip.gemm_bf16(A, W, M, N, K, want_g, bias=bias, gamma=gamma)
        for variant in (7, 0):
            lib.ufm_debug_set_gemm_variant(variant)
            for rep in range(3):
                got_b = torch.full((M, N), 3.0, device=DEV, dtype=torch.bfloat16)
                hip.gemm_bf16(A, W, M, N, K, got_b, bias=bias, act=1)
                assert torch.equal(got_b.view(torch.int16), want_b.view(torch.int16)), (variant, rep, "bias + GELU")
                got_g = torch.full((M, N), 3.0, device=DEV, dtype=torch.bfloat16)
                hip.gemm_bf16(A, W, M, N, K, got_g, bias=bias, gamma=gamma)
                assert torch.equal(got_g.view(torch.int16), want_g.view(torch.int16)), (variant, rep, "bias + scale")
    finally:
        lib.ufm_debug_set_gemm_variant(0)


def test_gemm_8phase_tile_heights_with_row_tables(hip):
    """The row-remapped epilogue (residual table indexed modulo a period, output rows skipping one slot per group -- the
    patch-embed / view-embedding forms) under every 8-phase tile height: bitwise the 128x128 kernel's result."""
    lib = hip.lib()
    M, N, K, period = 4 * 2738, 768, 1024, 2738
    A = bf16r(rnd(M, K, seed=1)).to(DEV).bfloat16()
    W = bf16r(rnd(N, K, seed=2, scale=K**-0.5)).to(DEV).bfloat16()
    bias, table = rnd(N, seed=3, scale=0.1).to(DEV), rnd(period, N, seed=5).to(DEV)
    group = 1369
    try:
        want = {}
        for variant, rows in ((1, 0), (4, 256), (4, 224), (4, 192), (4, 160), (0, 0)):
            lib.ufm_debug_set_gemm_variant(variant)
            lib.ufm_debug_set_gemm_tile_rows(rows)
            a = torch.full((M, N), 3.0, device=DEV)
            hip.gemm_bf16(A, W, M, N, K, a, bias=bias, res=table, res_row_mod=period)
            b = torch.full((M + M // group, N), 3.0, device=DEV)
            hip.gemm_bf16(A, W, M, N, K, b, bias=bias, out_row_group=group)
            if variant == 1:
                want = dict(a=a, b=b)
                assert torch.all(b[0 :: group + 1] == 3.0)
            else:
                assert torch.equal(a, want["a"]) and torch.equal(b, want["b"]), (variant, rows)
    finally:
        lib.ufm_debug_set_gemm_variant(0)
        lib.ufm_debug_set_gemm_tile_rows(0)


@pytest.mark.parametrize("variant", [1, 4])
def test_gemm_gelu_epilogue_accuracy(hip, variant):
    """The bf16-output GELU epilogue (gelu_bf16_x4: relu(x) - |x| 2^-g(|x|), one transcendental per value) against
    the exact erf form: after rounding to bf16 the result must equal bf16(exact) except for rare 1-ulp flips at
    rounding boundaries, over the whole bf16 grid in [-9, 9] incl. the negative tail."""
    lib = hip.lib()
    xs = torch.arange(-9.0, 9.0, 1.0 / 64).to(torch.bfloat16).unique().float()  # bf16-representable inputs
    M, N, K = xs.numel(), 256, 128
    A = torch.zeros(M, K)
    A[:, 0] = xs
    W = torch.zeros(N, K)
    W[:, 0] = 1.0
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    lib.ufm_debug_set_gemm_variant(variant)
    try:
        hip.gemm_bf16(A.to(DEV).bfloat16(), W.to(DEV).bfloat16(), M, N, K, out, act=1)
    finally:
        lib.ufm_debug_set_gemm_variant(0)
    got = out.cpu()
    want = F.gelu(xs.double()).to(torch.bfloat16)[:, None].expand(M, N)
    ulps = (got.view(torch.int16).int() - want.contiguous().view(torch.int16).int()).abs()
    fit = (xs >= -6.0)[:, None].expand(M, N)  # below -6 the kernel holds Phi at Phi(-6) = 1e-9: |gelu| < 1e-8 either way
    assert ulps[fit].max().item() <= 1, ulps[fit].max().item()
    assert (ulps[fit] == 0).float().mean().item() >= 0.99
    assert (got.float() - want.float())[~fit].abs().max().item() <= 2e-8


@pytest.mark.parametrize("variant,flags", [(1, 0), (4, 0), (5, 0), (1, 8)])
@pytest.mark.parametrize("M,N,K,out_bf16", [(256 * 70 + 13, 768, 192, True), (256 * 64 + 200, 1024, 640, False)])
def test_gemm_large_tile_variants(hip, variant, flags, M, N, K, out_bf16):
    """Every kernel choice of ufm_gemm_bf16 (128x128, 256x256 8-phase, hybrid split; flags 8 = the 128x128 kernel's direct
    epilogue) against the same fp64 statement, ragged M, short and long K (prologue/epilogue of the DMA ring)."""
    lib = hip.lib()
    lib.ufm_debug_set_gemm_flags(flags)
    A = bf16r(rnd(M, K, seed=1))
    W = bf16r(rnd(N, K, seed=2, scale=K**-0.5))
    bias, gamma, res = rnd(N, seed=3, scale=0.1), 1 + rnd(N, seed=4, scale=0.1), rnd(M, N, seed=5)
    ref = (A.double() @ W.double().T + bias.double()) * gamma.double() + res.double()
    out = torch.zeros(M, N, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=DEV)
    lib.ufm_debug_set_gemm_variant(variant)
    try:
        hip.gemm_bf16(A.to(DEV).bfloat16(), W.to(DEV).bfloat16(), M, N, K, out, bias=bias.to(DEV), gamma=gamma.to(DEV), res=res.to(DEV))
    finally:
        lib.ufm_debug_set_gemm_variant(0)
        lib.ufm_debug_set_gemm_flags(0)
    err = (out.float().cpu().double() - ref).abs().max().item()
    assert err <= (3e-2 if out_bf16 else 3e-4) * max(1.0, ref.abs().max().item()), err
    assert lib.ufm_debug_set_gemm_variant(3) != 0  # retired variants are rejected, not silently mapped


def test_gemm_in_place_residual(hip):
    M, N, K = 200, 128, 64
    A, W = bf16r(rnd(M, K, seed=1)), bf16r(rnd(N, K, seed=2))
    x = rnd(M, N, seed=3)
    xd = x.to(DEV).clone()
    hip.gemm_bf16(A.to(DEV).bfloat16(), W.to(DEV).bfloat16(), M, N, K, xd, res=xd)
    assert (xd.cpu() - (x + A @ W.T)).abs().max() < 1e-3


def test_gemm_rejects_bad_shapes(hip):
    A = torch.zeros(64, 96, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        hip.gemm_bf16(A, A, 64, 128, 96, torch.zeros(64, 128, device=DEV))


# ----------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("D", [128, 192, 256, 768, 1024, 2048])
@pytest.mark.parametrize("out_bf16", [False, True])
def test_layernorm(hip, D, out_bf16):
    rows = 37
    x = rnd(rows + 5, D, seed=1, scale=3.0) + 0.5
    w, b = 1 + rnd(D, seed=2, scale=0.1), rnd(D, seed=3, scale=0.1)
    idx = torch.randperm(rows + 5, generator=torch.Generator().manual_seed(0))[:rows].int()
    ref = F.layer_norm(x[idx.long()], (D,), w, b, 1e-6)
    out = torch.empty(rows, D, device=DEV, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    hip.layernorm(x.to(DEV), D, idx.to(DEV), rows, D, w.to(DEV), b.to(DEV), 1e-6, out)
    err = (out.float().cpu() - ref).abs().max().item()
    assert err <= (3e-2 if out_bf16 else 2e-5)


@pytest.mark.parametrize("D", [128, 768, 1024])
@pytest.mark.parametrize("out_kind", ["bf16", "f32", "split"])
@pytest.mark.parametrize("use_gamma", [False, True])
def test_add_layernorm(hip, D, out_kind, use_gamma):
    """ufm_add_layernorm: x += gamma * branch (fp32 math on the bf16 branch, x written back), then LayerNorm(x)."""
    rows = 301
    x = rnd(rows, D, seed=1, scale=2.0)
    br = bf16r(rnd(rows, D, seed=2))
    gamma = 1 + rnd(D, seed=3, scale=0.2) if use_gamma else None
    w, b = 1 + rnd(D, seed=4, scale=0.1), rnd(D, seed=5, scale=0.1)
    x_new = x + (gamma * br if gamma is not None else br)  # fp32, mul then add: what the kernel does (-ffp-contract=off)
    ref = F.layer_norm(x_new.double(), (D,), w.double(), b.double(), 1e-6)
    xd = x.to(DEV).clone()
    if out_kind == "split":
        out = torch.zeros(2, rows, D, device=DEV, dtype=torch.bfloat16)
    else:
        out = torch.zeros(rows, D, device=DEV, dtype=torch.bfloat16 if out_kind == "bf16" else torch.float32)
    hip.add_layernorm(xd, D, br.to(DEV).bfloat16(), gamma.to(DEV) if gamma is not None else None, rows, D, w.to(DEV), b.to(DEV), 1e-6, out, split=out_kind == "split")
    assert torch.equal(xd.cpu(), x_new), "the residual stream must hold exactly x + gamma * branch (fp32)"
    got = (out[0].float() + out[1].float()) if out_kind == "split" else out.float()
    err = (got.cpu().double() - ref).abs().max().item()
    assert err <= (3e-2 if out_kind == "bf16" else 4e-5), err  # split output: 2^-17 relative on |y| <~ 4


# ----------------------------------------------------------------------------- attention
def attn_ref(qkv, B, N, H, scale):
    q, k, v = qkv.double().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    p = torch.softmax((q @ k.transpose(-1, -2)) * scale, dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B * N, H * 64)


@pytest.mark.parametrize("B,N,H", [(1, 17, 1), (2, 100, 2), (2, 1370, 2), (1, 2738, 3), (1, 64, 1), (1, 129, 1)])
def test_attention_bf16(hip, B, N, H):
    qkv = bf16r(rnd(B * N, 3 * H * 64, seed=N, scale=1.5))
    ref = attn_ref(qkv, B, N, H, 0.125)
    out = torch.zeros(B * N, H * 64, device=DEV, dtype=torch.bfloat16)
    hip.attention(qkv.to(DEV).bfloat16(), out, B, N, H, 0.125)
    err = (out.float().cpu().double() - ref).abs().max().item()
    assert err <= 2e-2, err  # P and O are rounded to bf16 (8 bits): |O| <~ 1


@pytest.mark.parametrize("B,N,H", [(1, 17, 1), (2, 100, 2), (2, 1370, 2), (1, 2738, 3), (1, 64, 1), (1, 129, 1)])
@pytest.mark.parametrize("variant", [0, 1])
def test_attention_bf16_prescaled_log2_kernel(hip, B, N, H, variant):
    """scale == 0 path: Q columns pre-multiplied by scale*log2(e) (what the QKV GEMM epilogue does).
    variant 0 = 64-rows-per-wave kernel with 4 waves per workgroup (default), 1 = the same with 2 waves per workgroup."""
    hip.lib().ufm_debug_set_attn_variant(variant)
    qkv = rnd(B * N, 3 * H * 64, seed=N, scale=1.5)
    qkv_b = bf16r(qkv)
    ref = attn_ref(qkv_b, B, N, H, 0.125)
    pre = qkv.clone().reshape(B * N, 3, H * 64)
    pre[:, 0] = bf16r(pre[:, 0]) * (0.125 * 1.4426950408889634)  # same values the epilogue would round once
    pre = bf16r(pre.reshape(B * N, -1))
    pre.reshape(B * N, 3, H * 64)[:, 1:] = qkv_b.reshape(B * N, 3, H * 64)[:, 1:]
    out = torch.zeros(B * N, H * 64, device=DEV, dtype=torch.bfloat16)
    try:
        hip.attention(pre.to(DEV).bfloat16(), out, B, N, H, 0.0)
    finally:
        hip.lib().ufm_debug_set_attn_variant(0)
    err = (out.float().cpu().double() - ref).abs().max().item()
    assert err <= 3e-2, err  # + one extra bf16 rounding of the already-rounded test Q (not present in the fused pipeline)


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("growth", [3.0, 30.0])
def test_attention_bf16_prescaled_deferred_rescale_branches(hip, growth, variant):
    """Late keys whose scores exceed the running reference by less / more than the deferral threshold (rule 26):
    both the 'keep the reference' and the 'move the reference' paths must give the same softmax."""
    B, N, H = 1, 300, 1
    c = 0.125 * 1.4426950408889634
    qkv = bf16r(rnd(N, 192, seed=3, scale=0.5))
    qkv[250, 64:128] = bf16r(qkv[5, :64] * growth)   # key 250 spikes for query 5 (and partly for others)
    qkv[100, 64:128] = bf16r(qkv[40, :64] * growth)
    ref = attn_ref(qkv, B, N, H, 0.125)
    pre = qkv.clone()
    pre[:, :64] = bf16r(pre[:, :64] * c)
    refp = attn_ref(torch.cat([pre[:, :64] / c, pre[:, 64:]], 1), B, N, H, 0.125)  # exact statement of the pre-rounded problem
    out = torch.zeros(N, 64, device=DEV, dtype=torch.bfloat16)
    hip.lib().ufm_debug_set_attn_variant(variant)
    try:
        hip.attention(pre.to(DEV).bfloat16(), out, B, N, H, 0.0)
    finally:
        hip.lib().ufm_debug_set_attn_variant(0)
    assert (out.float().cpu().double() - refp).abs().max().item() <= 2e-2
    assert (out.float().cpu().double() - ref).abs().max().item() <= 6e-2


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("growth", [40.0, 400.0])
def test_attention_bf16_pw_reference_moves(hip, growth, variant):
    """The 64-rows-per-wave kernel keeps NO running maximum: the reference set at the first key tile only moves when a
    tile's sum of 2^(s - m_ref) passes 2^64 (or is inf).  Spike keys in the first / a middle / the ragged last tile, for
    queries of q-block A, q-block B and another wave, at growths that give finite-but-huge sums (40) and inf (400):
    every row must still match the fp64 softmax (rule 26: full-tensor reference, inputs that FORCE the branch)."""
    B, N, H = 1, 300, 1
    c = 0.125 * 1.4426950408889634
    qkv = bf16r(rnd(N, 192, seed=11, scale=0.5))
    for key, qrow in ((3, 7), (70, 5), (100, 40), (131, 100), (250, 5), (290, 40), (299, 170), (200, 299)):
        qkv[key, 64:128] = bf16r(qkv[qrow, :64] * growth)
    pre = qkv.clone()
    pre[:, :64] = bf16r(pre[:, :64] * c)
    refp = attn_ref(torch.cat([pre[:, :64] / c, pre[:, 64:]], 1), B, N, H, 0.125)
    out = torch.zeros(N, 64, device=DEV, dtype=torch.bfloat16)
    hip.lib().ufm_debug_set_attn_variant(variant)
    try:
        for _ in range(3):
            hip.attention(pre.to(DEV).bfloat16(), out, B, N, H, 0.0)
            got = out.float().cpu().double()
            assert torch.isfinite(got).all()
            assert (got - refp).abs().max().item() <= 2e-2
    finally:
        hip.lib().ufm_debug_set_attn_variant(0)


@pytest.mark.parametrize("variant", [0, 1])
def test_attention_bf16_pw_repeatable_full_size(hip, variant):
    """Race screen for the LDS-DMA ring (counted waits + one barrier per key tile): UFM-Base shapes, repeated launches
    must agree bit for bit, and with the independent scale > 0 kernel (un-scaled Q) to bf16 rounding."""
    lib = hip.lib()
    for B, N, H in ((2, 1370, 16), (1, 2738, 12)):
        raw = rnd(B * N, 3 * H * 64, seed=N, scale=1.0)
        want = torch.zeros(B * N, H * 64, device=DEV, dtype=torch.bfloat16)
        hip.attention(raw.to(DEV).bfloat16(), want, B, N, H, 0.125)
        pre = raw.clone()
        pre[:, : H * 64] = bf16r(raw[:, : H * 64]) * (0.125 * 1.4426950408889634)
        qkv = pre.to(DEV).bfloat16()
        lib.ufm_debug_set_attn_variant(variant)
        try:
            first = None
            for rep in range(5):
                got = torch.full((B * N, H * 64), 3.0, device=DEV, dtype=torch.bfloat16)
                hip.attention(qkv, got, B, N, H, 0.0)
                if first is None:
                    first = got.clone()
                    assert (got.float() - want.float()).abs().max().item() <= 6e-2  # bf16 ulps at |O| ~ 4 + the second rounding of Q
                assert torch.equal(got.view(torch.int16), first.view(torch.int16)), rep
        finally:
            lib.ufm_debug_set_attn_variant(0)


def test_attention_bf16_pw_repeatable_beside_another_stream(hip):
    """The engine runs two micro-batches on two HIP streams, so every kernel shares the chip with the other stream's
    kernels: workgroups start late, staggered and with a cold instruction cache.  The encoder-shaped attention launch must
    stay bitwise repeatable while a second stream runs the qkv GEMM (the pairing that exposed a missing barrier between the
    K(0) fragment reads and the K(2) LDS-DMA into the same ring buffer)."""
    import threading

    B, N, H, D = 8, 1370, 12, 768
    qkv = rnd(B * N, 3 * H * 64, seed=5, scale=1.0).to(DEV).bfloat16()
    ref = torch.zeros(B * N, H * 64, device=DEV, dtype=torch.bfloat16)
    hip.attention(qkv, ref, B, N, H, 0.0)
    torch.cuda.synchronize()
    A = rnd(B * N, D, seed=6, scale=1.0).to(DEV).bfloat16()
    Wt = rnd(3 * D, D, seed=7, scale=0.03).to(DEV).bfloat16()
    gout = torch.empty(B * N, 3 * D, device=DEV, dtype=torch.bfloat16)
    side = torch.cuda.Stream()
    stop = []

    def load():
        with torch.cuda.stream(side):
            while not stop:
                for _ in range(10):
                    hip.gemm_bf16(A, Wt, B * N, 3 * D, D, gout)
                side.synchronize()

    t = threading.Thread(target=load)
    t.start()
    try:
        for rep in range(60):
            got = torch.zeros_like(ref)
            hip.attention(qkv, got, B, N, H, 0.0)
            torch.cuda.synchronize()
            assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), rep
    finally:
        stop.append(1)
        t.join()


def test_attention_bf16_spike_forces_rescale(hip):
    """A late key with a huge score forces the online-softmax rescale branch (rule 26)."""
    B, N, H = 1, 300, 1
    qkv = bf16r(rnd(N, 192, seed=3, scale=0.5))
    qkv[:, :64] = bf16r(qkv[:, :64])
    qkv[250, 64:128] = bf16r(qkv[5, :64] * 30)  # key 250 aligned with query 5
    ref = attn_ref(qkv, B, N, H, 0.125)
    out = torch.zeros(N, 64, device=DEV, dtype=torch.bfloat16)
    hip.attention(qkv.to(DEV).bfloat16(), out, B, N, H, 0.125)
    assert (out.float().cpu().double() - ref).abs().max().item() <= 2e-2


@pytest.mark.parametrize("B,N,H", [(1, 17, 1), (2, 200, 2), (1, 1370, 1)])
def test_attention_f32(hip, B, N, H):
    qkv = rnd(B * N, 3 * H * 64, seed=N, scale=1.5)
    ref = attn_ref(qkv, B, N, H, 0.125)
    out = torch.zeros(B * N, H * 64, device=DEV)
    hip.attention(qkv.to(DEV), out, B, N, H, 0.125)
    assert (out.cpu().double() - ref).abs().max().item() <= 2e-5


# ----------------------------------------------------------------------------- conv (fp32 MFMA)
def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize(
    "B,H,W,Cin,Cout,k,stride,pad,relu_in,act,nres",
    [
        (2, 9, 11, 32, 32, 3, 1, 1, False, 0, 0),
        (1, 37, 37, 64, 128, 3, 1, 1, True, 0, 2),
        (2, 37, 37, 96, 64, 3, 2, 1, False, 0, 0),
        (1, 20, 13, 128, 96, 1, 1, 0, False, 2, 1),
        (1, 30, 30, 256, 256, 3, 1, 1, True, 0, 1),
    ],
)
def test_conv2d(hip, B, H, W, Cin, Cout, k, stride, pad, relu_in, act, nres):
    x = rnd(B, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, k, k, seed=2, scale=(Cin * k * k) ** -0.5)
    b = rnd(Cout, seed=3, scale=0.1)
    ref = F.conv2d(F.relu(x) if relu_in else x, w, b, stride=stride, padding=pad)
    if act == 2:
        ref = F.relu(ref)
    res = [rnd(*ref.shape, seed=10 + i) for i in range(nres)]
    for r in res:
        ref = ref + r
    Ho, Wo = ref.shape[2:]
    out = torch.zeros(B, Ho, Wo, Cout, device=DEV)
    zero = torch.zeros(64, device=DEV)
    resd = [nhwc(r).to(DEV) for r in res] + [None, None]
    hip.conv2d(nhwc(x).to(DEV), B, H, W, Cin, w.permute(0, 2, 3, 1).contiguous().to(DEV), Cout, k, k, stride, pad, out, zero,
               relu_in=relu_in, bias=b.to(DEV), act=act, res1=resd[0], res2=resd[1])
    err = (out.cpu().permute(0, 3, 1, 2) - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("s,Cin,Co", [(4, 32, 32), (2, 64, 48)])
def test_conv_transpose_as_shuffle(hip, s, Cin, Co):
    B, H, W = 2, 5, 7
    x = rnd(B, Cin, H, W, seed=1)
    wt = rnd(Cin, Co, s, s, seed=2, scale=Cin**-0.5)
    b = rnd(Co, seed=3, scale=0.1)
    ref = F.conv_transpose2d(x, wt, b, stride=s)
    wp = wt.permute(2, 3, 1, 0).reshape(s * s * Co, Cin).contiguous()
    out = torch.zeros(B, H * s, W * s, Co, device=DEV)
    zero = torch.zeros(64, device=DEV)
    hip.conv2d(nhwc(x).to(DEV), B, H, W, Cin, wp.to(DEV), s * s * Co, 1, 1, 1, 0, out, zero, bias=b.to(DEV), shuffle=s)
    assert (out.cpu().permute(0, 3, 1, 2) - ref).abs().max().item() <= 2e-5


def test_conv_as_dense_gemm(hip):
    M, K, N = 333, 128, 384
    a, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K**-0.5)
    out = torch.zeros(M, N, device=DEV)
    hip.conv2d(a.to(DEV), 1, 1, M, K, w.to(DEV), N, 1, 1, 1, 0, out, torch.zeros(64, device=DEV), act=1)
    assert (out.cpu() - F.gelu(a @ w.T)).abs().max().item() <= 2e-5


# ----------------------------------------------------------------------------- pointwise
@pytest.mark.parametrize("Ho,Wo,crop", [(20, 26, None), (37, 37, None), (14, 18, (13, 17))])
def test_upsample(hip, Ho, Wo, crop):
    B, H, W, Cc = 2, 7, 9, 8
    x = rnd(B, Cc, H, W, seed=1)
    ref = F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=True)
    ch, cw = crop if crop else (Ho, Wo)
    ref = ref[:, :, :ch, :cw]
    out = torch.zeros(B, ch, cw, Cc, device=DEV)
    hip.upsample_bilinear(nhwc(x).to(DEV), B, H, W, Cc, out, Ho, Wo, *(crop or (0, 0)))
    assert (out.cpu().permute(0, 3, 1, 2) - ref).abs().max().item() <= 1e-5


def test_upsample_scale2_matches_scale_factor(hip):
    x = rnd(1, 4, 19, 19, seed=2)
    ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    out = torch.zeros(1, 38, 38, 4, device=DEV)
    hip.upsample_bilinear(nhwc(x).to(DEV), 1, 19, 19, 4, out, 38, 38)
    assert (out.cpu().permute(0, 3, 1, 2) - ref).abs().max().item() <= 1e-5


def test_head_tail(hip):
    B, HW, Cin = 2, 35, 32
    x, w, b = rnd(B * HW, Cin, seed=1), rnd(3, Cin, seed=2), rnd(3, seed=3)
    y = x @ w.T + b
    out = torch.zeros(B, 3, HW, device=DEV)
    logits = torch.zeros(B, 3, HW, device=DEV)
    hip.head_tail(x.to(DEV), B * HW, HW, Cin, w.to(DEV), b.to(DEV), 3, [0, 0, 1], [2.0, 0.5, 1.0], [0.1, -0.2, 0.0], out, logits)
    yy = y.reshape(B, HW, 3).permute(0, 2, 1)
    assert (out.cpu()[:, 0] - (yy[:, 0] * 2.0 + 0.1)).abs().max() <= 1e-5
    assert (out.cpu()[:, 1] - (yy[:, 1] * 0.5 - 0.2)).abs().max() <= 1e-5
    assert (out.cpu()[:, 2] - torch.sigmoid(yy[:, 2])).abs().max() <= 1e-6
    assert (logits.cpu()[:, 2] - yy[:, 2]).abs().max() <= 1e-5


def test_patchify_u8_and_f32(hip):
    from oracle import uniception_ref as U

    B, H, W, P = 2, 28, 42, 14
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, generator=g)
    n = U.IMAGE_NORMALIZATION_DICT["dinov2"]
    norm = (img.permute(0, 3, 1, 2).float() / 255.0 - n.mean.view(1, 3, 1, 1)) / n.std.view(1, 3, 1, 1)
    ref = F.unfold(norm, kernel_size=P, stride=P).transpose(1, 2).reshape(-1, 3 * P * P)  # (c,i,j) column order
    out = torch.full((ref.shape[0], 640), 5.0, device=DEV)
    hip.patchify(img.to(DEV), 0, B, H, W, P, n.std.tolist(), n.mean.tolist(), out, 640)
    assert torch.equal(out.cpu()[:, :588], ref)  # same fp32 expression -> bit exact
    assert torch.all(out.cpu()[:, 588:] == 0)
    out2 = torch.zeros(ref.shape[0], 640, device=DEV, dtype=torch.bfloat16)
    hip.patchify(norm.contiguous().to(DEV), 1, B, H, W, P, [1, 1, 1], [0, 0, 0], out2, 640)
    assert torch.equal(out2.cpu()[:, :588], ref.bfloat16())


@pytest.mark.parametrize("src_hw,dst_hw", [((75, 100), (42, 56)), ((30, 40), (56, 70)), ((56, 56), (56, 56)), ((1080, 607), (518, 518))])
def test_resize_antialias(hip, src_hw, dst_hw):
    B = 1
    x = rnd(B, 3, *src_hw, seed=4)
    ref = F.interpolate(x, size=dst_hw, mode="bilinear", align_corners=False, antialias=True)
    out = torch.zeros(B, 3, *dst_hw, device=DEV)
    tmp = torch.zeros(B * 3 * src_hw[0] * dst_hw[1], device=DEV)
    hip.resize_antialias(x.to(DEV), 1, B, src_hw[0], src_hw[1], [1, 1, 1], [0, 0, 0], out, dst_hw[0], dst_hw[1], tmp)
    err = (out.cpu() - ref).abs().max().item()
    assert err <= (0.0 if src_hw == dst_hw else 5e-6), err


@pytest.mark.parametrize("name", ["unmap_full.npz", "unmap_crop.npz"])
def test_unmap_vs_reference_goldens(hip, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name))
    fl, ch = torch.from_numpy(g["flow_in"]), torch.from_numpy(g["chan_in"])
    B, _, h, w = fl.shape
    H0, W0 = (int(v) for v in g["shape0"])
    fo = torch.full((B, 2, H0, W0), 9.0, device=DEV)
    fv = torch.zeros(B, H0, W0, dtype=torch.uint8, device=DEV)
    hip.unmap_flow(fl.to(DEV), B, h, w, g["rep0"].tolist(), g["src0"].tolist(), g["src1"].tolist(), H0, W0, fo, fv)
    # coordinates are O(100) px in fp32 (ulp 7.6e-6); two independent fp32 evaluations differ by a few ulp
    assert np.abs(fo.cpu().numpy() - g["flow_out"]).max() <= 5e-5
    assert np.array_equal(fv.cpu().numpy().astype(bool), g["flow_valid"])
    co = torch.full((B, 3, H0, W0), 9.0, device=DEV)
    hip.unmap_channels(ch.to(DEV), B, 3, h, w, g["rep0"].tolist(), g["src0"].tolist(), H0, W0, co)
    assert np.array_equal(co.cpu().numpy(), g["chan_out"])


@pytest.mark.parametrize("name", ["refine_p5.npz", "refine_p3.npz"])
def test_refine_vs_reference_goldens(hip, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name))
    flow, feats = torch.from_numpy(g["flow"]), torch.from_numpy(g["feats"])
    B, _, H, W = flow.shape
    P = int(g["patch"])
    res = torch.zeros(B, 2, H, W, device=DEV)
    logp = torch.zeros(B, H, W, P, P, device=DEV)
    hip.refine(flow.to(DEV), feats.to(DEV), B, feats.shape[1], H, W, P, float(g["temperature"]), torch.from_numpy(g["bias"]).to(DEV), res, logp)
    assert np.abs(res.cpu().numpy() - g["residual"]).max() <= 2e-4
    assert np.abs(logp.cpu().numpy() - g["log_softmax"]).max() <= 2e-4


def test_pixel_shuffle_and_misc(hip):
    B, gh, gw, Cc, p = 2, 3, 4, 5, 14
    x = rnd(B * gh * gw, Cc * p * p, seed=1)
    ref = F.pixel_shuffle(x.reshape(B, gh, gw, Cc * p * p).permute(0, 3, 1, 2), p)
    out = torch.zeros(B, Cc, gh * p, gw * p, device=DEV)
    hip.pixel_shuffle_planar(x.to(DEV), B, gh, gw, Cc, p, out)
    assert torch.equal(out.cpu(), ref)
    # split-format input (the bf16x3 classification head's output): value = hi + lo
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    out2 = torch.zeros_like(out)
    hip.pixel_shuffle_planar(torch.stack([hi, lo]).contiguous().to(DEV), B, gh, gw, Cc, p, out2, split=True)
    ref2 = F.pixel_shuffle((hi.float() + lo.float()).reshape(B, gh, gw, Cc * p * p).permute(0, 3, 1, 2), p)
    assert torch.equal(out2.cpu(), ref2)
    a, b = rnd(1024, seed=2), rnd(1024, seed=3)
    o = torch.zeros(1024, device=DEV)
    hip.add_f32(a.to(DEV), b.to(DEV), o)
    assert torch.equal(o.cpu(), a + b)
    ob = torch.zeros(1024, device=DEV, dtype=torch.bfloat16)
    hip.cast_bf16(a.to(DEV), ob)
    assert torch.equal(ob.cpu(), a.bfloat16())
    rows, D, grp = 8, 16, 4
    t, tab = rnd(rows, D, seed=5), rnd(grp, D, seed=6)
    oo = torch.full((rows + rows // grp, D), 3.0, device=DEV)
    hip.add_rows(t.to(DEV), D, tab.to(DEV), grp, oo, D, grp, rows, D)
    idx = torch.arange(rows)
    orow = (idx // grp) * (grp + 1) + 1 + idx % grp
    assert torch.equal(oo.cpu()[orow], t + tab[idx % grp])
    fr = torch.zeros(10, D, device=DEV)
    hip.fill_rows(fr, D, 2, 5, tab[0].contiguous().to(DEV), D)
    assert torch.equal(fr.cpu()[0], tab[0]) and torch.equal(fr.cpu()[5], tab[0]) and fr.cpu()[1:5].abs().sum() == 0


# ----------------------------------------------------------------------------- bf16x3 split-precision path
def split(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return torch.stack([hi, lo], 0).contiguous()


def unsplit(s):
    return s[0].float() + s[1].float()


@pytest.mark.parametrize(
    "B,H,W,Cin,Cout,k,stride,pad,relu_in,act,nres",
    [
        (2, 9, 11, 32, 32, 3, 1, 1, False, 0, 0),
        (1, 37, 37, 64, 128, 3, 1, 1, True, 0, 2),
        (2, 37, 37, 96, 64, 3, 2, 1, False, 0, 0),
        (1, 20, 13, 128, 96, 1, 1, 0, False, 2, 1),
        (1, 30, 30, 256, 256, 3, 1, 1, True, 0, 1),
        (4, 128, 128, 32, 128, 3, 1, 1, True, 0, 1),   # >= 400 blocks: the 128x128 tile path
        (4, 128, 128, 32, 64, 3, 1, 1, False, 2, 0),   # 128x64 tile path
    ],
)
def test_conv2d_bf16x3(hip, B, H, W, Cin, Cout, k, stride, pad, relu_in, act, nres):
    x = rnd(B, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, k, k, seed=2, scale=(Cin * k * k) ** -0.5)
    b = rnd(Cout, seed=3, scale=0.1)
    ref = F.conv2d((F.relu(x) if relu_in else x).double(), w.double(), b.double(), stride=stride, padding=pad)
    if act == 2:
        ref = F.relu(ref)
    res = [rnd(*ref.shape, seed=10 + i) for i in range(nres)]
    for r in res:
        ref = ref + r.double()
    Ho, Wo = ref.shape[2:]
    out = torch.zeros(2, B, Ho, Wo, Cout, device=DEV, dtype=torch.bfloat16)
    zero = torch.zeros(256, device=DEV)
    resd = [split(nhwc(r)).to(DEV) for r in res] + [None, None]
    hip.conv2d_x3(split(nhwc(x)).to(DEV), B, H, W, Cin, split(w.permute(0, 2, 3, 1).contiguous()).to(DEV), Cout, k, k, stride, pad, out, zero,
                  relu_in=relu_in, bias=b.to(DEV), act=act, res1=resd[0], res2=resd[1])
    got = unsplit(out.cpu()).permute(0, 3, 1, 2).double()
    err = (got - ref).abs().max().item()
    # split inputs carry 2^-17 relative error, the dropped lo*lo term 2^-16 per product, the split store 2^-17
    assert err <= 4e-5 * max(1.0, ref.abs().max().item()), err


# ---- the cross-attention info-sharing variant's kernels: two-source attention, RoPE-2D (standalone and fused into the GEMM) ----
def cross_attn_ref(q, k, v, B, Nq, Nk, H, scale):
    qh = q.double().reshape(B, Nq, H, 64).permute(0, 2, 1, 3)
    kh = k.double().reshape(B, Nk, H, 64).permute(0, 2, 1, 3)
    vh = v.double().reshape(B, Nk, H, 64).permute(0, 2, 1, 3)
    p = torch.softmax((qh @ kh.transpose(-1, -2)) * scale, dim=-1)
    return (p @ vh).permute(0, 2, 1, 3).reshape(B * Nq, H * 64)


@pytest.mark.parametrize("B,Nq,Nk,H", [(1, 17, 40, 1), (2, 100, 37, 2), (2, 300, 1369, 2), (1, 1369, 130, 3), (3, 64, 64, 1)])
@pytest.mark.parametrize("fmt", ["bf16", "f32", "x3"])
def test_cross_attention_two_sources(hip, B, Nq, Nk, H, fmt):
    """ufm_cross_attention_*: queries [B*Nq][ldq] against keys / values of ANOTHER buffer [B*Nk][ldkv], Nq != Nk, K and V as
    column blocks of one K|V projection output (ldkv = 2 H 64), Q inside a wider buffer (ldq > H 64)."""
    D = H * 64
    qbuf = rnd(B * Nq, D + 64, seed=Nq, scale=1.5)       # queries in columns [0, D) of a wider buffer
    kv = rnd(B * Nk, 2 * D, seed=Nk + 1, scale=1.5)
    if fmt == "bf16":
        qbuf, kv = bf16r(qbuf), bf16r(kv)
    elif fmt == "x3":
        qs, kvs = split(qbuf), split(kv)
        qbuf, kv = unsplit(qs), unsplit(kvs)
    ref = cross_attn_ref(qbuf[:, :D], kv[:, :D], kv[:, D:], B, Nq, Nk, H, 0.125)
    if fmt == "x3":
        qd, kvd = qs.to(DEV), kvs.to(DEV)
        out = torch.zeros(2, B * Nq, D, device=DEV, dtype=torch.bfloat16)
        hip.cross_attention(qd[0][:, :D], D + 64, kvd[0][:, :D], kvd[0][:, D:], 2 * D, out[0], D, B, Nq, Nk, H, 0.125, hip.BF16X2)
        got, tol = unsplit(out.cpu()).double(), 1e-4
    else:
        dt = torch.bfloat16 if fmt == "bf16" else torch.float32
        qd, kvd = qbuf.to(DEV).to(dt), kv.to(DEV).to(dt)
        out = torch.zeros(B * Nq, D, device=DEV, dtype=dt)
        hip.cross_attention(qd[:, :D], D + 64, kvd[:, :D], kvd[:, D:], 2 * D, out, D, B, Nq, Nk, H, 0.125, hip.BF16 if fmt == "bf16" else hip.F32)
        got, tol = out.float().cpu().double(), (2e-2 if fmt == "bf16" else 2e-5)
    err = (got - ref).abs().max().item()
    assert err <= tol, err


@pytest.mark.parametrize("B,Nq,Nk,H", [(1, 17, 40, 1), (2, 100, 37, 2), (2, 300, 1369, 2), (1, 1369, 130, 3), (3, 64, 64, 1), (2, 257, 513, 2)])
def test_cross_attention_two_sources_prescaled_on_the_persistent_kernel(hip, B, Nq, Nk, H):
    """ufm_cross_attention_bf16 with scale == 0 (q pre-scaled by softmax_scale * log2 e, as the Q projection's epilogue leaves it):
    the persistent LDS-DMA kernel of attention_bf16_pw.hip with separate q and k / v sources, Nq != Nk, ragged last key tile and
    ragged last query block, q inside a wider buffer."""
    D = H * 64
    qbuf = bf16r(rnd(B * Nq, D + 64, seed=Nq, scale=1.5))
    kv = bf16r(rnd(B * Nk, 2 * D, seed=Nk + 1, scale=1.5))
    ref = cross_attn_ref(qbuf[:, :D], kv[:, :D], kv[:, D:], B, Nq, Nk, H, 0.125)
    pre = qbuf.clone()
    pre[:, :D] = bf16r(pre[:, :D] * (0.125 * 1.4426950408889634))
    qd, kvd = pre.to(DEV).bfloat16(), kv.to(DEV).bfloat16()
    out = torch.zeros(B * Nq, D, device=DEV, dtype=torch.bfloat16)
    hip.cross_attention(qd[:, :D], D + 64, kvd[:, :D], kvd[:, D:], 2 * D, out, D, B, Nq, Nk, H, 0.0, hip.BF16)
    err = (out.float().cpu().double() - ref).abs().max().item()
    assert err <= 3e-2, err  # bf16 P and O + the test's extra rounding of the pre-scaled q


@pytest.mark.parametrize("B,Np,H", [(2, 1369, 2), (3, 100, 1), (1, 37, 3)])
def test_attention_strided_view1_queries_are_bitwise_the_joint_attention_rows(hip, B, Np, H):
    """The last joint-attention layer decodes view 1 only (/root/reference/uniflowmatch/models/ufm.py:637-641): its view-1 queries
    (the first Np of a pair's 2 Np rows, q_batch_rows = 2 Np) against ALL 2 Np keys through ufm_attention_bf16_strided must equal
    the corresponding rows of the full joint self-attention bit for bit (same kernel, same key-tile order per query row --
    whenever the 256-row query blocks of the two launches align, i.e. always for the leading rows of a batch item)."""
    N, D = 2 * Np, H * 64
    qkv = rnd(B * N, 3 * D, seed=Np, scale=1.5)
    qkv[:, :D] *= 0.125 * 1.4426950408889634  # pre-scaled q: scores of a few units, so no wave takes the (wave-wide) deferred-rescale path
    qkv = qkv.to(DEV).bfloat16()
    full = torch.zeros(B * N, D, device=DEV, dtype=torch.bfloat16)
    hip.attention(qkv, full, B, N, H, 0.0)
    half = torch.full((B * Np, D), 7.0, device=DEV, dtype=torch.bfloat16)
    hip.attention_strided(qkv[:, :D], 3 * D, N, qkv[:, D : 2 * D], qkv[:, 2 * D :], 3 * D, N, half, D, Np, B, Np, N, H)
    want = full.view(B, 2, Np, D)[:, 0].reshape(B * Np, D)
    assert torch.equal(half, want)
    # and into a strided output (rows of a wider buffer, batch items 2 Np rows apart)
    wide = torch.zeros(B * N, D + 64, device=DEV, dtype=torch.bfloat16)
    hip.attention_strided(qkv[:, :D], 3 * D, N, qkv[:, D : 2 * D], qkv[:, 2 * D :], 3 * D, N, wide, D + 64, N, B, Np, N, H)
    assert torch.equal(wide.view(B, 2, Np, D + 64)[:, 0, :, :D].reshape(B * Np, D), want)
    assert bool((wide.view(B, 2, Np, D + 64)[:, 1] == 0).all()) and bool((wide[:, D:] == 0).all())  # nothing else was written


def test_gather_rows_f32(hip):
    x = rnd(50, 72, seed=3).to(DEV)
    idx = torch.tensor([49, 0, 7, 7, 13, 48], dtype=torch.int32, device=DEV)
    out = torch.full((6, 80), 5.0, device=DEV)
    hip.gather_rows(x[:, :64], 72, idx, 6, 64, out, 80)
    assert torch.equal(out[:, :64], x[idx.long(), :64]) and bool((out[:, 64:] == 5.0).all())


def test_attention_strided_rejects_bad_arguments(hip):
    q = torch.zeros(128, 192, device=DEV, dtype=torch.bfloat16)
    o = torch.zeros(64, 64, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="batch strides"):
        hip.attention_strided(q[:, :64], 192, 32, q[:, 64:128], q[:, 128:], 192, 128, o, 64, 64, 1, 64, 128, 1)
    with pytest.raises(RuntimeError, match="densely packed"):
        hip.attention_strided(q[:, :64], 192, 128, q[:, 64:128], q[:, 128:], 192, 128, o, 64, 64, 1, 64, 128, 1, scale=0.125)


def rope_ref(x, gh, gw, H, freq=100.0):
    """The oracle's CroCo RoPE2D on a (rows = B*gh*gw, H*64) projection output."""
    from oracle.uniception_ref import RoPE2D, grid_positions

    Np = gh * gw
    B = x.shape[0] // Np
    t = x.reshape(B, Np, H, 64).permute(0, 2, 1, 3)
    return RoPE2D(freq)(t, grid_positions(B, gh, gw)).permute(0, 2, 1, 3).reshape(B * Np, H * 64)


def rope_tables(gh, gw, freq=100.0):
    """Host tables as the engine builds them, via the engine's own routine on a stub."""
    import types

    from ufm_amd.engine import Engine

    stub = types.SimpleNamespace(_tables={}, rope_freq=freq, dev=torch.device(DEV))
    return Engine._rope_tables(stub, gh, gw)


@pytest.mark.parametrize("fmt", ["f32", "bf16", "x3"])
def test_rope2d_standalone_matches_the_oracle(hip, fmt):
    gh, gw, H, B = 5, 7, 3, 2
    Np, D = gh * gw, H * 64
    x = rnd(B * Np, 3 * D, seed=7)                      # a QKV buffer: rotate q and k (columns [0, 2D)), leave v alone
    cos, sin = rope_tables(gh, gw)
    if fmt == "bf16":
        x = bf16r(x)
    if fmt == "x3":
        xs = split(x)
        x = unsplit(xs)
        buf = xs.to(DEV)
    else:
        buf = x.to(DEV).to(torch.bfloat16 if fmt == "bf16" else torch.float32)
    want = torch.cat([rope_ref(x[:, :D], gh, gw, H), rope_ref(x[:, D : 2 * D], gh, gw, H), x[:, 2 * D :]], dim=1)
    hip.rope2d(buf, B * Np, 3 * D, 0, 2 * D, cos, sin, Np)
    got = unsplit(buf.cpu()) if fmt == "x3" else buf.float().cpu()
    err = (got - want).abs().max().item()
    assert err <= (3e-2 if fmt == "bf16" else 2e-5 if fmt == "x3" else 2e-6), err
    assert torch.equal(got[:, 2 * D :], x[:, 2 * D :]) or fmt == "x3"  # the v columns are untouched


@pytest.mark.parametrize("M_mult,N,K,rope_cols", [(3, 384, 128, 256), (40, 3072, 1024, 2048), (9, 256, 192, 256)])
def test_gemm_fused_rope_matches_linear_then_rope(hip, M_mult, N, K, rope_cols):
    """ufm_gemm_bf16_rope ("fused QKV+RoPE"): the rotation runs on the fp32 accumulator (+ bias, x gamma) before the bf16
    rounding; against fp64 Linear -> oracle RoPE2D, on the 128x128 and the 8-phase kernels, with a ragged last row tile."""
    gh, gw = 5, 7
    Np = gh * gw
    M = M_mult * Np
    A = bf16r(rnd(M, K, seed=1))
    W = bf16r(rnd(N, K, seed=2, scale=K**-0.5))
    bias, gamma = rnd(N, seed=3, scale=0.1), 1 + rnd(N, seed=4, scale=0.1)
    lin = ((A.double() @ W.double().T + bias.double()) * gamma.double()).float()
    H = rope_cols // 64
    want = torch.cat([rope_ref(lin[:, :rope_cols], gh, gw, H), lin[:, rope_cols:]], dim=1).double()
    cos, sin = rope_tables(gh, gw)
    out = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    hip.gemm_bf16(A.to(DEV).bfloat16(), W.to(DEV).bfloat16(), M, N, K, out, bias=bias.to(DEV), gamma=gamma.to(DEV), rope=(cos, sin, Np, rope_cols))
    err = (out.float().cpu().double() - want).abs().max().item()
    assert err <= 2e-2 * max(1.0, want.abs().max().item()), err
    # and the un-rotated columns are bitwise those of the plain GEMM
    plain = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    hip.gemm_bf16(A.to(DEV).bfloat16(), W.to(DEV).bfloat16(), M, N, K, plain, bias=bias.to(DEV), gamma=gamma.to(DEV))
    if rope_cols < N:
        assert torch.equal(out[:, rope_cols:], plain[:, rope_cols:])
    with pytest.raises(RuntimeError, match="plain bf16 output"):
        hip.gemm_bf16(A.to(DEV).bfloat16(), W.to(DEV).bfloat16(), M, N, K, torch.zeros(M, N, device=DEV), rope=(cos, sin, Np, rope_cols))


# ---- the moge_conv head's kernels: replicate-padding convolution, GroupNorm, uv channels, bilinear (align_corners=False) ----
@pytest.mark.parametrize("fmt", ["f32", "x3"])
@pytest.mark.parametrize("B,H,W,Cin,Cout,big", [(2, 9, 11, 32, 32, False), (1, 30, 30, 256, 256, True), (1, 37, 21, 64, 128, False)])
def test_conv3x3_replicate_padding(hip, fmt, B, H, W, Cin, Cout, big):
    """padding_mode="replicate" (relu_in flag bit 1) on the fp32 and the bf16x3 conv kernels, incl. the 8-phase 256x256 tile."""
    x = rnd(B, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=(Cin * 9) ** -0.5)
    b = rnd(Cout, seed=3, scale=0.1)
    ref = F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode="replicate"), w.double(), b.double())
    zero = torch.zeros(256, device=DEV)
    if fmt == "f32":
        out = torch.zeros(B, H, W, Cout, device=DEV)
        hip.conv2d(nhwc(x).to(DEV), B, H, W, Cin, w.permute(0, 2, 3, 1).contiguous().to(DEV), Cout, 3, 3, 1, 1, out, zero, bias=b.to(DEV), replicate=True)
        got, tol = out.cpu().permute(0, 3, 1, 2).double(), 2e-5
    else:
        if big:
            hip.lib().ufm_debug_set_conv_variant(2)  # force the 8-phase kernel
        out = torch.zeros(2, B, H, W, Cout, device=DEV, dtype=torch.bfloat16)
        hip.conv2d_x3(split(nhwc(x)).to(DEV), B, H, W, Cin, split(w.permute(0, 2, 3, 1).contiguous()).to(DEV), Cout, 3, 3, 1, 1, out, zero, bias=b.to(DEV), replicate=True)
        hip.lib().ufm_debug_set_conv_variant(0)
        got, tol = unsplit(out.cpu()).permute(0, 3, 1, 2).double(), 4e-5
    err = (got - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("fmt", ["f32", "x3"])
@pytest.mark.parametrize("B,H,W,C,G,relu", [(2, 9, 11, 64, 2, True), (1, 70, 33, 256, 8, True), (3, 5, 5, 32, 1, False), (1, 40, 40, 96, 3, True)])
def test_group_norm_nhwc(hip, fmt, B, H, W, C, G, relu):
    x = rnd(B, C, H, W, seed=1, scale=2.0) + 0.5
    w, b = 1 + rnd(C, seed=2, scale=0.1), rnd(C, seed=3, scale=0.1)
    ref = F.group_norm(x.double(), G, w.double(), b.double(), 1e-5)
    if relu:
        ref = F.relu(ref)
    xin = nhwc(x)
    if fmt == "x3":
        xs = split(xin)
        ref = F.group_norm(unsplit(xs).permute(0, 3, 1, 2).double(), G, w.double(), b.double(), 1e-5)
        ref = F.relu(ref) if relu else ref
        buf, out = xs.to(DEV), torch.zeros(2, B, H, W, C, device=DEV, dtype=torch.bfloat16)
    else:
        buf, out = xin.to(DEV), torch.zeros(B, H, W, C, device=DEV)
    ws = torch.zeros(hip.group_norm_ws_floats(B, H * W, G), device=DEV)
    hip.group_norm(buf, B, H * W, C, G, w.to(DEV), b.to(DEV), 1e-5, relu, out, ws)
    got = (unsplit(out.cpu()) if fmt == "x3" else out.cpu()).permute(0, 3, 1, 2).double()
    err = (got - ref).abs().max().item()
    assert err <= 3e-5, err
    again = torch.zeros_like(out)
    hip.group_norm(buf, B, H * W, C, G, w.to(DEV), b.to(DEV), 1e-5, relu, again, ws)
    assert torch.equal(again, out)  # deterministic statistics


@pytest.mark.parametrize("fmt", ["f32", "x3"])
def test_fill_uv_and_bilinear_resize_into_concat_buffer(hip, fmt):
    """torch.cat([F.interpolate(x, (H, W), bilinear, align_corners=False), uv], dim=1) as MoGe builds it, in a 32-channel-padded
    NHWC buffer: ufm_resize_bilinear_nhwc writes the x slot, ufm_fill_uv_nhwc the two uv channels and the zero padding."""
    from oracle.uniception_ref import normalized_view_plane_uv

    B, h, w, C, H, W = 2, 16, 24, 32, 45, 70
    ldc = 64
    x = rnd(B, C, h, w, seed=4)
    xin = nhwc(x)
    if fmt == "x3":
        xs = split(xin)
        x = unsplit(xs).permute(0, 3, 1, 2)
        buf, out = xs.to(DEV), torch.full((2, B, H, W, ldc), 7.0, device=DEV, dtype=torch.bfloat16)
    else:
        buf, out = xin.to(DEV), torch.full((B, H, W, ldc), 7.0, device=DEV)
    want_x = F.interpolate(x, (H, W), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    uv = normalized_view_plane_uv(W, H, W / H)
    hip.resize_bilinear(buf, B, h, w, C, C, out, H, W, ldc, 0)
    hip.fill_uv(out, B, H, W, ldc, C, W / H)
    got = unsplit(out.cpu()) if fmt == "x3" else out.cpu()
    assert (got[..., :C] - want_x).abs().max().item() <= (2e-5 if fmt == "x3" else 2e-6)
    assert (got[..., C : C + 2] - uv.unsqueeze(0)).abs().max().item() <= (8e-6 if fmt == "x3" else 1e-6)  # (hi, lo) carries 2^-17 relative
    assert got[..., C + 2 :].abs().max().item() == 0.0


# ---- numerics "precise": the transformer's Linear layers and attention on the split format ----
@pytest.mark.parametrize(
    "M,N,K,act,use_gamma,use_res,split_out",
    [
        (300, 256, 128, 0, False, False, True),
        (300, 64, 96, 1, True, False, True),            # GELU (exact erf), 128x64 tiles
        (257, 128, 64, 0, True, True, False),           # LayerScale + fp32 residual in place
        (2 * 1370, 3072, 1024, 0, True, False, True),   # QKV shape: 8-phase 256x256 tiles + remainder
        (1370, 4096, 1024, 1, False, False, True),      # fc1 shape: GELU + split store on whole 8-phase tiles (the grouped store-only epilogue)
        (2 * 1370, 1024, 4096, 0, True, True, False),   # fc2 shape
        (8 * 1369 + 5, 768, 768, 0, False, True, False),
    ],
)
def test_gemm_bf16x3(hip, M, N, K, act, use_gamma, use_res, split_out):
    A = rnd(M, K, seed=1)
    W = rnd(N, K, seed=2, scale=K**-0.5)
    bias = rnd(N, seed=3, scale=0.1)
    gamma = 1 + rnd(N, seed=4, scale=0.1) if use_gamma else None
    res = rnd(M, N, seed=5) if use_res else None
    As, Ws = split(A), split(W)
    ref = unsplit(As).double() @ unsplit(Ws).double().T + bias.double()  # the kernel sees the split operands
    if act == 1:
        ref = F.gelu(ref)
    if gamma is not None:
        ref = ref * gamma.double()
    if res is not None:
        ref = ref + res.double()
    zero = torch.zeros(256, device=DEV)
    if split_out:
        out = torch.full((2, M, N), 7.0, device=DEV, dtype=torch.bfloat16)
        hip.gemm_x3(As.to(DEV), Ws.to(DEV), M, N, K, out, zero, bias=bias.to(DEV), act=act, gamma=gamma.to(DEV) if gamma is not None else None)
        got = unsplit(out.cpu()).double()
    else:
        out = res.to(DEV).clone() if res is not None else torch.full((M, N), 7.0, device=DEV)
        hip.gemm_x3(As.to(DEV), Ws.to(DEV), M, N, K, out, zero, bias=bias.to(DEV), act=act, gamma=gamma.to(DEV) if gamma is not None else None,
                    res=out if res is not None else None)
        got = out.cpu().double()
    err = (got - ref).abs().max().item()
    # dropped lo*lo terms 2^-16 per product (random signs), fp32 accumulation over K, split store 2^-17
    assert err <= 4e-5 * max(1.0, ref.abs().max().item()), err
    if use_res:  # round 5: the grouped residual loads of the epilogue against the serial per-pass read-out (variant + 16), bit for bit
        lib = hip.lib()
        try:
            lib.ufm_debug_set_conv_variant(16)
            out2 = res.to(DEV).clone()
            hip.gemm_x3(As.to(DEV), Ws.to(DEV), M, N, K, out2, zero, bias=bias.to(DEV), act=act, gamma=gamma.to(DEV) if gamma is not None else None, res=out2)
        finally:
            lib.ufm_debug_set_conv_variant(0)
        assert torch.equal(out2, out)
    if split_out:  # ... and of the store-only split form (mode 4; with the GELU: fc1), likewise against the serial read-out
        lib = hip.lib()
        try:
            lib.ufm_debug_set_conv_variant(16)
            out2 = torch.full((2, M, N), 7.0, device=DEV, dtype=torch.bfloat16)
            hip.gemm_x3(As.to(DEV), Ws.to(DEV), M, N, K, out2, zero, bias=bias.to(DEV), act=act, gamma=gamma.to(DEV) if gamma is not None else None)
        finally:
            lib.ufm_debug_set_conv_variant(0)
        assert torch.equal(out2.view(torch.int16), out.view(torch.int16))


@pytest.mark.parametrize(
    "M,N,K,act,use_gamma,use_res,split_out",
    [
        (300, 256, 64, 0, False, False, True),            # two K-tiles: prologue + tail waits only
        (2 * 1370, 3072, 1024, 0, True, False, True),     # encoder QKV (Q pre-scale as gamma)
        (1370, 4096, 1024, 1, False, False, True),        # encoder fc1: GELU + split store
        (8 * 1369 + 5, 2304, 768, 0, True, False, True),  # info-sharing QKV, ragged rows
        (2738, 3072, 768, 1, False, False, True),         # info-sharing fc1
        (513, 1024, 4096, 0, True, True, False),          # fc2 shape: fp32 read-modify-write (not fed by a LayerNorm in the engine; the kernel supports it)
    ],
)
def test_gemm_bf16x3_interleaved_operands_are_bitwise_the_planar_form(hip, M, N, K, act, use_gamma, use_res, split_out):
    """Round 6: ufm_gemm_bf16x3_il reads A and W INTERLEAVED per 32-channel chunk ([rows][K / 32][hi 32 | lo 32]: every
    LDS-DMA row of the 8-phase loop one whole 128-byte line) -- same K order, same three products per fragment pair, same epilogues as
    ufm_gemm_bf16x3 on the planar planes: bit-identical outputs, at every tile height (the Linear layers of
    /root/reference/uniflowmatch/models/ufm.py:187,193 in numerics "precise")."""
    lib = hip.lib()
    A = split(rnd(M, K, seed=1)).to(DEV)
    W = split(rnd(N, K, seed=2, scale=K**-0.5)).to(DEV)
    Ai, Wi = hip.interleave_split(A), hip.interleave_split(W)
    assert Ai.shape == (M, K // 32, 2, 32) and torch.equal(Ai[5, 1, 1], A[1, 5, 32:64])
    bias = rnd(N, seed=3, scale=0.1).to(DEV)
    gamma = (1 + rnd(N, seed=4, scale=0.1)).to(DEV) if use_gamma else None
    res = rnd(M, N, seed=5).to(DEV) if use_res else None
    zero = torch.zeros(256, device=DEV)

    def run(fn, a, w):
        if split_out:
            o = torch.full((2, M, N), 7.0, device=DEV, dtype=torch.bfloat16)
            fn(a, w, M, N, K, o, zero, bias=bias, act=act, gamma=gamma)
            return o.view(torch.int16).clone()
        o = res.clone()
        fn(a, w, M, N, K, o, zero, bias=bias, act=act, gamma=gamma, res=o)
        return o.view(torch.int32).clone()

    want = run(hip.gemm_x3, A, W)
    try:
        for variant in (0, 2 | (5 << 8), 2 | (6 << 8), 2 | (7 << 8), 2 | (8 << 8), 16):  # (16: the serial per-pass epilogue)
            assert lib.ufm_debug_set_conv_variant(variant) == 0
            got = run(hip.gemm_x3_il, Ai, Wi)
            assert torch.equal(got, want), (variant, int((got != want).sum()))
            if split_out:  # ... and with the split OUTPUT interleaved too (UFM_BF16X2_IL: fc1 -> fc2): the same values at other addresses
                o = torch.full((M, N // 32, 2, 32), 7.0, device=DEV, dtype=torch.bfloat16)
                hip.gemm_x3_il(Ai, Wi, M, N, K, o, zero, bias=bias, act=act, gamma=gamma)
                assert torch.equal(o.view(torch.int16), hip.interleave_split(want.view(torch.bfloat16)).view(torch.int16)), variant
    finally:
        lib.ufm_debug_set_conv_variant(0)


@pytest.mark.parametrize("B,N,H", [(2, 1370, 16), (1, 2738, 12), (3, 77, 4)])
def test_attention_bf16x3_interleaved_output(hip, B, N, H):
    """ufm_attention_bf16x3_il: O stored [B N][H 64 / 32][hi 32 | lo 32] (the proj Linear's operand in numerics "precise") -- bitwise the
    planar output of ufm_attention_bf16x3, re-laid (SDPA under /root/reference/uniflowmatch/models/base.py:272-274)."""
    qkv = split(rnd(B * N, 3 * H * 64, seed=1)).to(DEV)
    planar = torch.full((2, B * N, H * 64), 7.0, device=DEV, dtype=torch.bfloat16)
    hip.attention_x3(qkv, planar, B, N, H, 0.125)
    il = torch.full((B * N, H * 2, 2, 32), 7.0, device=DEV, dtype=torch.bfloat16)
    hip.attention_x3(qkv, il, B, N, H, 0.125, out_interleaved=True)
    assert torch.equal(il.view(torch.int16), hip.interleave_split(planar).view(torch.int16))
    # scale = 0: the Q columns pre-scaled by softmax_scale * log2(e) (what the engine's QKV epilogue does in "precise"), no multiply per score
    # in the kernel -- the same softmax against the fp64 statement on the ORIGINAL q
    q0 = unsplit(qkv.cpu())
    ref = attn_ref(q0, B, N, H, 0.125)
    pre = q0.clone()
    pre[:, : H * 64] *= 0.125 * 1.4426950408889634
    il2 = torch.full((B * N, H * 2, 2, 32), 7.0, device=DEV, dtype=torch.bfloat16)
    hip.attention_x3(split(pre).to(DEV), il2, B, N, H, 0.0, out_interleaved=True)
    got = unsplit(il2.permute(2, 0, 1, 3).reshape(2, B * N, H * 64).cpu()).double()
    assert (got - ref).abs().max().item() <= 1e-4


@pytest.mark.parametrize("rows,D", [(37, 256), (1370, 1024), (2738, 768)])
def test_layernorm_interleaved_split_output(hip, rows, D):
    """ufm_layernorm with out_dtype UFM_BF16X2_IL: the same (hi, lo) values as the planar split output, laid out [row][D / 32][hi 32 | lo 32]."""
    x = rnd(rows, D, seed=1).to(DEV)
    w, b = (1 + rnd(D, seed=2, scale=0.1)).to(DEV), rnd(D, seed=3, scale=0.1).to(DEV)
    planar = torch.full((2, rows, D), 7.0, device=DEV, dtype=torch.bfloat16)
    hip.layernorm(x, D, None, rows, D, w, b, 1e-6, planar, split=True)
    il = torch.full((rows, D // 32, 2, 32), 7.0, device=DEV, dtype=torch.bfloat16)
    hip.layernorm(x, D, None, rows, D, w, b, 1e-6, il, split=True, interleaved=True)
    assert torch.equal(il.view(torch.int16), hip.interleave_split(planar).view(torch.int16))


@pytest.mark.parametrize("M,N,K,use_res,act", [(512, 256, 128, True, 0), (512, 256, 128, False, 0), (512, 256, 128, False, 2)])
def test_gemm_bf16x3_nan_in_is_nan_out_in_the_grouped_epilogue(hip, M, N, K, use_res, act):
    """the grouped (round 5) epilogue applied 'no activation' as fmaxf(v, -inf) -- fmaxf returns its non-NaN operand, so a NaN
    accumulator became -inf (fp32 residual stream) or (hi = -inf, lo = NaN) in the split store, where the serial per-pass read-out
    (variant + 16) kept the NaN.  Now ReLU / identity are one signed-integer max on the bit pattern in BOTH forms: a NaN row stays NaN,
    and the two forms agree bit for bit on NaN data too (act 0: fp32 read-modify-write and split store; act 2 = ReLU: the two forms agree)."""
    lib = hip.lib()
    A = rnd(M, K, seed=1)
    A[7, 3] = float("nan")      # row 7 of the output is NaN in every column
    A[300, 100] = float("inf")  # row 300: +-inf products of both signs -> NaN / inf mix
    W = rnd(N, K, seed=2, scale=K**-0.5)
    As, Ws = split(A).to(DEV), split(W).to(DEV)
    bias, zero = rnd(N, seed=3, scale=0.1).to(DEV), torch.zeros(256, device=DEV)
    res = rnd(M, N, seed=5).to(DEV)
    outs = []
    try:
        for variant in (0, 16):
            lib.ufm_debug_set_conv_variant(variant)
            if use_res:
                o = res.clone()
                hip.gemm_x3(As, Ws, M, N, K, o, zero, bias=bias, res=o)
                outs.append(o.view(torch.int32).clone())
            else:
                o = torch.full((2, M, N), 7.0, device=DEV, dtype=torch.bfloat16)
                hip.gemm_x3(As, Ws, M, N, K, o, zero, bias=bias, act=act)
                outs.append(o.view(torch.int16).clone())
    finally:
        lib.ufm_debug_set_conv_variant(0)
    assert torch.equal(outs[0], outs[1])
    val = outs[0].view(torch.float32) if use_res else unsplit(outs[0].view(torch.bfloat16))
    if act != 2:  # (under ReLU the MFMA's NaNs -- sign bit set on gfx950 -- clamp to 0 in both forms, as fmaxf(NaN, 0) = 0 always did)
        assert torch.isnan(val[7]).all() and torch.isnan(val[300]).any()
    keep = torch.ones(M, dtype=torch.bool)
    keep[7] = keep[300] = False
    assert torch.isfinite(val[keep.to(val.device)]).all()


def test_split_format_gelu_epilogue_against_fp64_gelu(hip):
    """The GELU of every split-format epilogue (conv_x3_common.h: gelu_erf_fast -- Abramowitz-Stegun 7.1.26 on v_rcp_f32 /
    v_exp2_f32) measured directly: a 32 x 32 identity weight and every bf16 value of [-8, 8] as the operand, so the
    accumulator IS x (hi * 1, lo = 0: exact) and the fp32 output is gelu_erf_fast(x) alone.  This GELU serves the precise-mode
    fc1 and -- since round 3 -- the classification-head MLP in "fast" and "parity_x3heads" too (the DPT heads have no GELU)."""
    bits = torch.arange(0, 1 << 16, dtype=torch.int32)
    vals = (bits << 16).view(torch.float32)
    vals = vals[torch.isfinite(vals) & (vals.abs() <= 8.0)]
    n = (vals.numel() // 32) * 32
    x = vals[:n].reshape(-1, 32).contiguous()
    M = x.shape[0]
    A = torch.stack([x.bfloat16(), torch.zeros_like(x).bfloat16()])
    assert torch.equal(A[0].float(), x)  # every operand is a bf16 value: the hi plane alone carries it
    W = torch.stack([torch.eye(32).bfloat16(), torch.zeros(32, 32).bfloat16()])
    out = torch.full((M, 32), 7.0, device=DEV)
    hip.gemm_x3(A.to(DEV), W.to(DEV), M, 32, 32, out, torch.zeros(256, device=DEV), act=1)
    ref = F.gelu(x.double())
    err = (out.cpu().double() - ref).abs()
    rel = err / ref.abs().clamp_min(1e-300)
    # Measured on MI355X (tools/lab/gelu_probe.py): max |err| 4.4e-7 (at x = 3.27: the erfc polynomial's 1.5e-7 times x / 2, plus
    # the approximate v_rcp / v_exp2); relative error <= 6e-7 for x >= -1, 5.6e-5 on [-3, -1] and an ABSOLUTE 2.6e-7 below -3,
    # where gelu itself is < 4e-3 (the formula bounds the absolute error; it does not keep relative accuracy in the negative tail,
    # unlike gelu_bf16_x4 of the bf16 GEMM).  Against the 2^-17 = 7.6e-6 relative resolution of the split store: an order below it.
    assert err.max().item() <= 1e-6, err.max().item()
    assert rel[x >= -1.0].max().item() <= 2e-6 and rel[(x >= -3.0) & (x < -1.0)].max().item() <= 2e-4


def test_gemm_bf16x3_rejects_bad_arguments(hip):
    z = torch.zeros(256, device=DEV)
    a, w = torch.zeros(2, 64, 48, device=DEV, dtype=torch.bfloat16), torch.zeros(2, 64, 48, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="multiple of 32"):
        hip.gemm_x3(a, w, 64, 64, 48, torch.zeros(2, 64, 64, device=DEV, dtype=torch.bfloat16), z)
    a, w = torch.zeros(2, 64, 64, device=DEV, dtype=torch.bfloat16), torch.zeros(2, 64, 64, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="fp32 output"):
        hip.gemm_x3(a, w, 64, 64, 64, torch.zeros(2, 64, 64, device=DEV, dtype=torch.bfloat16), z, res=torch.zeros(64, 64, device=DEV))


@pytest.mark.parametrize("B,N,H", [(1, 17, 1), (2, 100, 2), (2, 1370, 2), (1, 2738, 3), (1, 64, 1), (1, 129, 1)])
def test_attention_bf16x3(hip, B, N, H):
    qkv = rnd(B * N, 3 * H * 64, seed=N, scale=1.5)
    qs = split(qkv)
    ref = attn_ref(unsplit(qs), B, N, H, 0.125)
    out = torch.zeros(2, B * N, H * 64, device=DEV, dtype=torch.bfloat16)
    hip.attention_x3(qs.to(DEV), out, B, N, H, 0.125)
    err = (unsplit(out.cpu()).double() - ref).abs().max().item()
    # |O| <~ 1; a score carries ~2^-17 * sum|q||k| (|q|,|k| ~ 1.5: up to 1e-4 in the exponent at d = 64), P and O are split to
    # 2^-17; measured 4.4e-5 at N = 1370 (the single-pass bf16 kernel: 2e-2)
    assert err <= 1e-4, err


@pytest.mark.parametrize("B,N,H", [(2, 1370, 2), (1, 2738, 1), (3, 200, 2), (1, 64, 1), (1, 37, 1), (2, 129, 3), (1, 64 * 5 + 1, 2)])
def test_attention_bf16x3_round5_kernel_is_bitwise_the_round1_kernel(hip, B, N, H):
    """attention_bf16x3_pw.hip (round 5: LDS-DMA rings, QK^T of tile t + 1 interleaved with the softmax of tile t) performs, per
    accumulator, exactly the MFMA sequence and the softmax operations of attention_bf16x3.hip in its order: the outputs must agree
    BIT FOR BIT -- repeated launches (a DMA that lands late, or a ring stage re-filled early, shows as a rare wrong tile), one tile
    only (N <= 64), odd tile counts, a ragged last key tile, a ragged last query block; with moving maxima (large scores)."""
    lib = hip.lib()
    qkv = rnd(B * N, 3 * H * 64, seed=N, scale=1.5)
    qkv[N // 2, H * 64 : H * 64 + 64] = qkv[N // 3, 0:64] * 9.0  # one key aligned with one query: a late, large maximum move
    qs = split(qkv).to(DEV)
    try:
        lib.ufm_debug_set_attn_variant(2)  # the round-1 kernel
        want = torch.full((2, B * N, H * 64), 7.0, device=DEV, dtype=torch.bfloat16)
        hip.attention_x3(qs, want, B, N, H, 0.125)
        for variant in (8, 12):  # the running-maximum form (bit 3; round 6's default is the fixed softmax reference): four waves per workgroup, eight (A/B)
            lib.ufm_debug_set_attn_variant(variant)
            for rep in range(4):
                got = torch.full((2, B * N, H * 64), 3.0, device=DEV, dtype=torch.bfloat16)
                hip.attention_x3(qs, got, B, N, H, 0.125)
                assert torch.equal(got.view(torch.int16), want.view(torch.int16)), (variant, rep)
        # round 6's default (fixed reference): the same softmax with other roundings -- close to the round-1 kernel, and repeatable bit for bit
        lib.ufm_debug_set_attn_variant(0)
        a = torch.full((2, B * N, H * 64), 3.0, device=DEV, dtype=torch.bfloat16)
        hip.attention_x3(qs, a, B, N, H, 0.125)
        ref = attn_ref(unsplit(qs.cpu()), B, N, H, 0.125)
        e_new, e_old = (unsplit(a.cpu()).double() - ref).abs().max().item(), (unsplit(want.cpu()).double() - ref).abs().max().item()
        assert e_new <= max(1e-4, 1.05 * e_old + 1e-6), (e_new, e_old)  # (the test's x 9 key is 2^233 above its row's first tile: the cold re-reference path; measured 1.22e-4 for BOTH kernels on it)
        for rep in range(3):
            b_ = torch.full((2, B * N, H * 64), 5.0, device=DEV, dtype=torch.bfloat16)
            hip.attention_x3(qs, b_, B, N, H, 0.125)
            assert torch.equal(a.view(torch.int16), b_.view(torch.int16)), rep
    finally:
        lib.ufm_debug_set_attn_variant(0)


@pytest.mark.parametrize("first_tile_high", [False, True])
def test_attention_bf16x3_fixed_reference_shift_path(hip, first_tile_high):
    """Round 6: the split-precision kernel's softmax reference is the row's maximum over its FIRST key tile.  (a) Later keys scoring far above it
    (here up to 2^150 times the first tile's weight): the cold path moves the reference up to that tile's maximum and redoes the tile, repeatedly; (b) a first tile far ABOVE everything
    else: later weights underflow towards zero, as they should.  Both against the fp64 softmax, every row (SDPA under
    /root/reference/uniflowmatch/models/base.py:272-274)."""
    B, N, H = 1, 64 * 7 + 5, 2
    qkv = rnd(B * N, 3 * H * 64, seed=21, scale=0.5)
    if first_tile_high:
        qkv[5, H * 64 : H * 64 + 64] = qkv[9, 0:64] * 40.0          # key 5 (tile 0) aligned with query 9: ~ +80 in the exponent, nothing later comes close
    else:
        for t in range(1, 6):
            qkv[t * 64 + 13, H * 64 : H * 64 + 64] = qkv[3, 0:64] * (12.0 * t)  # query 3: scores growing by ~24 nats per tile (2^35 per step)
    qs = split(qkv)
    ref = attn_ref(unsplit(qs), B, N, H, 0.125)
    out = torch.zeros(2, B * N, H * 64, device=DEV, dtype=torch.bfloat16)
    hip.attention_x3(qs.to(DEV), out, B, N, H, 0.125)
    got = unsplit(out.cpu()).double()
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item()
    assert err <= 1e-4, err


def test_attention_bf16x3_spike_moves_the_running_maximum(hip):
    """One key per 64-key tile with a growing score: every tile moves the running maximum (the rescale branch), then a long
    flat stretch where it never moves (the skipped-rescale branch); checked against the fp64 statement on every row."""
    B, N, H = 1, 64 * 9 + 7, 1
    qkv = rnd(B * N, 3 * 64, seed=11, scale=0.5)
    for t in range(5):
        qkv[t * 64 + 13, 64:128] = qkv[3, 0:64] * (4.0 + 6.0 * t)  # key t*64+13 aligned with query 3 (and friends)
    qs = split(qkv)
    ref = attn_ref(unsplit(qs), B, N, H, 0.125)
    out = torch.zeros(2, B * N, H * 64, device=DEV, dtype=torch.bfloat16)
    hip.attention_x3(qs.to(DEV), out, B, N, H, 0.125)
    err = (unsplit(out.cpu()).double() - ref).abs().max().item()
    assert err <= 1e-4, err


@pytest.mark.parametrize(
    "B,H,W,Cin,Cout,k,stride,pad,relu_in,act",
    [
        (1, 37, 37, 64, 128, 3, 1, 1, True, 0),     # 64x64 tiles
        (4, 128, 128, 32, 128, 3, 1, 1, False, 2),  # 128x128 tiles
        (4, 128, 128, 64, 64, 3, 1, 1, False, 2),   # 128x64 tiles
        (2, 64, 64, 128, 96, 1, 1, 0, False, 0),    # 128x32 tiles, 1x1
        (2, 9, 11, 96, 1024, 1, 1, 0, False, 0),    # Cout = 1024 (the 8-phase kernel is never used for passes = 1)
    ],
)
def test_conv2d_single_pass_is_a_bf16_convolution(hip, B, H, W, Cin, Cout, k, stride, pad, relu_in, act):
    """passes = 1 (the UNet under the reference's bf16 autocast, ufm.py:915-917): only the hi planes of the split operands
    enter the MFMAs, i.e. conv(bf16(x), bf16(w)) with fp32 accumulation -- compared with that statement in fp64; the lo planes
    of input and weight (filled with junk here) must not matter."""
    x = rnd(B, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, k, k, seed=2, scale=(Cin * k * k) ** -0.5)
    b = rnd(Cout, seed=3, scale=0.1)
    xb, wb = bf16r(x), bf16r(w)
    ref = F.conv2d((F.relu(xb) if relu_in else xb).double(), wb.double(), b.double(), stride=stride, padding=pad)
    if act == 2:
        ref = F.relu(ref)
    Ho, Wo = ref.shape[2:]
    xs, ws = split(nhwc(x)), split(w.permute(0, 2, 3, 1).contiguous())
    xs[1] = rnd(*xs[1].shape, seed=7).to(torch.bfloat16)  # junk in the lo planes
    ws[1] = rnd(*ws[1].shape, seed=8).to(torch.bfloat16)
    out = torch.zeros(2, B, Ho, Wo, Cout, device=DEV, dtype=torch.bfloat16)
    hip.conv2d_x3(xs.to(DEV), B, H, W, Cin, ws.to(DEV), Cout, k, k, stride, pad, out, torch.zeros(256, device=DEV), relu_in=relu_in, bias=b.to(DEV), act=act, passes=1)
    got = unsplit(out.cpu()).permute(0, 3, 1, 2).double()
    err = (got - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err  # fp32 accumulation order + the split store (2^-17)


@pytest.mark.parametrize(
    "B,H,W,Cin,Cout,k,stride,pad,relu_in,act,nres,shuffle",
    [
        (2, 37, 37, 64, 256, 3, 1, 1, True, 0, 2, 0),     # ragged M (2738 px), both residuals, ReLU on the input
        (1, 75, 61, 96, 256, 3, 2, 1, False, 2, 0, 0),    # stride 2, ReLU on the output
        (3, 23, 17, 32, 512, 1, 1, 0, False, 0, 1, 0),    # 1x1, Cin = 32: a single K-tile, below the 8-phase kernel's minimum -> falls back
        (3, 23, 17, 64, 512, 1, 1, 0, False, 0, 1, 0),    # 1x1, nt = 2 (prologue only + tail waits)
        (2, 9, 11, 96, 1024, 1, 1, 0, False, 0, 0, 2),    # ConvTranspose (pixel-shuffle store), Co = 256
        (8, 148, 148, 32, 256, 3, 1, 1, True, 0, 1, 0),   # 685 tiles, 9 K-tiles: auto = the 128-row kernels (fewer than 16 K-tiles)
        (4, 148, 148, 64, 256, 3, 1, 1, True, 0, 1, 0),   # 343 tiles, 18 K-tiles: auto = 1 whole round on the 8-phase kernel + 128-row rest (87 tiles < half a round)
        (5, 148, 148, 64, 256, 3, 1, 1, False, 0, 0, 0),  # 428 tiles: the last partial round (172 tiles >= half the chip) stays on the 8-phase kernel
        (2, 19, 19, 768, 256, 3, 1, 1, False, 0, 0, 0),   # 46 blocks of 64x64, 216 K-tiles: the deep ring incl. its drain
        (3, 41, 37, 64, 128, 3, 1, 1, True, 0, 1, 0),     # Cout = 128: the 512 px x 128 cout 8-phase layout (ragged M = 4551)
        (2, 30, 30, 96, 384, 3, 2, 1, False, 2, 0, 0),    # Cout = 384 = 3 x 128: three column tiles of the 512 x 128 layout, stride 2
        (1, 30, 30, 256, 64, 3, 1, 1, True, 0, 2, 0),     # 15 blocks of 64x64, 72 K-tiles, both residuals
        (1, 9, 7, 32, 64, 1, 1, 0, False, 0, 0, 0),       # a single K-tile (nk = 1 < ring depth)
        (4, 74, 74, 64, 256, 3, 1, 1, True, 0, 2, 0),     # whole 8-phase tiles with BOTH residuals: the grouped loads in two groups of 8 passes
        (8, 37, 37, 64, 256, 3, 1, 1, False, 2, 1, 0),    # one residual + output ReLU (the -inf / 0 clamp of the grouped path)
    ],
)
def test_conv2d_bf16x3_8phase_bit_identical(hip, B, H, W, Cin, Cout, k, stride, pad, relu_in, act, nres, shuffle):
    """The 256x256 8-phase kernel and the deep-ring (NS = 4) small-grid kernels do the same arithmetic in the same order
    as the plain 2-stage 128-row kernels: outputs must be BIT-identical (variant 3 = plain 2-stage only, 1 = 128-row
    kernels incl. the deep ring, 2 = 8-phase everywhere, 0 = the auto/hybrid choice)."""
    lib = hip.lib()
    x = split(nhwc(rnd(B, Cin, H, W, seed=1))).to(DEV)
    w = split(rnd(Cout, k, k, Cin, seed=2, scale=(Cin * k * k) ** -0.5)).to(DEV)
    b = rnd(Cout // (shuffle * shuffle) if shuffle else Cout, seed=3, scale=0.1).to(DEV)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    oshape = (2, B, Ho * shuffle, Wo * shuffle, Cout // (shuffle * shuffle)) if shuffle else (2, B, Ho, Wo, Cout)
    res = [split(rnd(B, Ho, Wo, Cout, seed=10 + i)).to(DEV) for i in range(nres)] + [None, None]
    zero = torch.zeros(256, device=DEV)
    outs = {}
    try:
        # 19 = 3 + 16: the plain 2-stage 128-row kernels with the SERIAL per-pass residual read-out of rounds 1-4: the baseline that
        # round 5's grouped-load epilogue (conv_x3_common.h) must reproduce bit for bit in every kernel
        # 2 | nf << 8: the 8-phase kernel pinned to tiles of 32 nf pixels (round 5: 160 / 192 / 224 rows; the second half of a wave's rows
        # has nf - 4 fragments, the epilogue's grouped loads cover 4 / 8 / 12 passes)
        # 4: the 256 px x 128 cout two-resident-workgroups kernel (round 5) wherever Cout % 128 == 0
        for variant in (19, 3, 1, 2, 0, 2 | (5 << 8), 2 | (6 << 8), 2 | (7 << 8), 4):
            lib.ufm_debug_set_conv_variant(variant)
            out = torch.full(oshape, 7.0, device=DEV, dtype=torch.bfloat16)
            orl = None if shuffle else torch.full(oshape, 7.0, device=DEV, dtype=torch.bfloat16)
            hip.conv2d_x3(x, B, H, W, Cin, w, Cout, k, k, stride, pad, out, zero, relu_in=relu_in, bias=b, act=act,
                          res1=res[0], res2=res[1], shuffle=shuffle, out_relu=orl)
            outs[variant] = (out.cpu(), None if shuffle else orl.cpu())
    finally:
        lib.ufm_debug_set_conv_variant(0)
    for v in (3, 1, 2, 0, 2 | (5 << 8), 2 | (6 << 8), 2 | (7 << 8), 4):
        assert torch.equal(outs[19][0].view(torch.int16), outs[v][0].view(torch.int16)), v
        if not shuffle:
            assert torch.equal(outs[19][1].view(torch.int16), outs[v][1].view(torch.int16)), v
    if not shuffle:  # the second output is relu(out), exactly (what relu_in=1 would apply by the sign of hi)
        assert torch.equal(unsplit(outs[3][1]), torch.relu(unsplit(outs[3][0])))


@pytest.mark.parametrize(
    "B,H,W,Cin,Cout,act,nres,groups",
    [
        (8, 148, 148, 256, 256, 0, 1, 1),   # the heads' dominant layer (RCU conv2 + skip): 685 tiles, 24 windows per tile pair of super-tiles
        (2, 148, 148, 96, 256, 0, 0, 1),    # Cin = 96: an ODD number of windows (9) -- the loop is entered at its second half
        (3, 74, 74, 192, 256, 2, 2, 1),     # W = 74: two row ends per 128-pixel wave row; both residuals; output ReLU
        (5, 37, 41, 64, 256, 0, 1, 1),      # non-square, W odd, H W = 1517: tiles straddle IMAGE boundaries (top / bottom rows voided per window)
        (7, 19, 33, 32, 256, 0, 0, 1),      # Cin = 32: three windows (the minimum); (H - 2) W = 561; ragged last tile
        (1, 64, 32, 64, 512, 0, 0, 1),      # W = 32 (the minimum: one row end per fragment), two cout tiles
        (4, 40, 48, 128, 256, 0, 1, 2),     # a grouped launch (the two DPT heads in one grid): per-group windows, ragged per-group tiles
        (1, 11, 32, 32, 256, 2, 0, 1),      # the eligibility boundary ((H - 2) W = 288 >= 272): ONE image of 352 pixels -- tile 1 is ragged and its windows run past the tensor's end
        (2, 11, 32, 64, 256, 0, 1, 1),      # the same map twice: an image boundary INSIDE tile 1 and the tensor's end inside tile 2
    ],
)
def test_conv2d_bf16x3_halo_bit_identical(hip, B, H, W, Cin, Cout, act, nres, groups):
    """Round 6: conv_bf16x3_halo.hip -- the 8-phase 3x3 kernel with the input staged once per FILTER ROW (a row-window halo tile) instead of
    once per tap, row borders voided when a window is staged (out-of-range buffer offsets write zeros to LDS), column borders zeroed in
    the fragment registers.  Same K order, same products: bit-identical to the gather form of the 8-phase kernel (variant HALO = 1:
    bit 5), at every tile height (nf 5..8), and to the plain 128-row kernels with the serial epilogue (variant 19), through the
    reference's DPT convolution shapes (/root/reference/uniflowmatch/models/ufm.py:243-289) and the awkward ones: odd window counts,
    tiles across image boundaries, W = 32, ragged tiles, groups."""
    lib = hip.lib()
    G = groups
    x = split(nhwc(rnd(G * B, Cin, H, W, seed=1))).to(DEV)
    w = split(rnd(G * Cout, 3, 3, Cin, seed=2, scale=(Cin * 9) ** -0.5)).to(DEV)
    b = rnd(G * Cout, seed=3, scale=0.1).to(DEV)
    oshape = (2, G * B, H, W, Cout)
    res = [split(rnd(G * B, H, W, Cout, seed=10 + i)).to(DEV) for i in range(nres)] + [None, None]
    zero = torch.zeros(256, device=DEV)
    outs = {}
    HALO_OFF = 1 << 5
    variants = (19, 2 | HALO_OFF, 2, 0, 2 | (5 << 8), 2 | (6 << 8), 2 | (7 << 8), 2 | (7 << 8) | HALO_OFF)
    # ... and every halo arm once more with the weights staged from their INTERLEAVED copy (ufm_conv_x3_register_interleaved_weights: whole
    # 128-byte DMA rows; keys 1000 + variant)
    w_il = hip.interleave_split(w.view(2, G * Cout, 9 * Cin))
    try:
        for wil in (False, True):
            if wil:
                assert lib.ufm_conv_x3_register_interleaved_weights(w.data_ptr(), w_il.data_ptr()) == 0
            for variant in variants:
                if wil and (variant == 19 or variant & HALO_OFF):
                    continue
                assert lib.ufm_debug_set_conv_variant(variant) == 0
                out = torch.full(oshape, 7.0, device=DEV, dtype=torch.bfloat16)
                orl = torch.full(oshape, 7.0, device=DEV, dtype=torch.bfloat16)
                hip.conv2d_x3(x, B, H, W, Cin, w, Cout, 3, 3, 1, 1, out, zero, bias=b, act=act, res1=res[0], res2=res[1], out_relu=orl, groups=G)
                outs[variant + (1000 if wil else 0)] = (out.view(torch.int16).clone(), orl.view(torch.int16).clone())
    finally:
        lib.ufm_debug_set_conv_variant(0)
        assert lib.ufm_conv_x3_register_interleaved_weights(w.data_ptr(), None) == 0
    for v in [k for k in outs if k != 19]:
        assert torch.equal(outs[19][0], outs[v][0]), (v, int((outs[19][0] != outs[v][0]).sum()))
        assert torch.equal(outs[19][1], outs[v][1]), v
    # ... and the numbers are a convolution: against the fp64 statement on the split operands (first group)
    xs, ws = unsplit(x.cpu())[:B].permute(0, 3, 1, 2).double(), unsplit(w.cpu())[:Cout].permute(0, 3, 1, 2).double()
    ref = F.conv2d(xs, ws, b[:Cout].cpu().double(), padding=1)
    if act == 2:
        ref = torch.relu(ref)
    for r in res[:nres]:
        ref = ref + unsplit(r.cpu())[:B].permute(0, 3, 1, 2).double()
    got = unsplit(outs[2][0].view(torch.bfloat16).cpu())[:B].permute(0, 3, 1, 2).double()
    assert (got - ref).abs().max().item() <= 4e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize(
    "B,H,W,Cin,Cout,k,stride,pad,act,nres,shuffle,shared",
    [
        (2, 37, 37, 256, 256, 3, 1, 1, 0, 2, 0, False),   # small grid: 64x64 tiles / deep ring
        (8, 37, 37, 256, 256, 3, 1, 1, 2, 1, 0, False),   # 128-row kernels or the 8-phase kernel on the doubled grid
        (4, 74, 74, 256, 256, 3, 1, 1, 0, 2, 0, False),   # 8-phase, partial round
        (5, 148, 148, 64, 256, 3, 1, 1, 0, 0, 0, False),  # 8-phase whole rounds + the hybrid rest launch (ragged last tile per group)
        (3, 19, 23, 128, 96, 1, 1, 0, 0, 0, 0, True),     # 1x1 projection of a SHARED input (the pyramid level both heads read)
        (2, 21, 21, 64, 64, 3, 2, 1, 0, 0, 0, False),     # stride 2
        (2, 9, 11, 64, 256, 1, 1, 0, 0, 0, 2, True),      # ConvTranspose (pixel shuffle), shared input
        (2, 40, 40, 256, 128, 3, 1, 1, 0, 0, 0, False),   # Cout = 128 (p_conv1's shape)
    ],
)
def test_conv2d_bf16x3_grouped_is_bitwise_the_separate_launches(hip, B, H, W, Cin, Cout, k, stride, pad, act, nres, shuffle, shared):
    """ufm_conv2d_nhwc_bf16x3_grouped: two convolutions of identical geometry (the two DPT heads' layers) in one grid must give,
    group by group, exactly the bits of two separate launches -- whatever kernel / tile shape / hybrid split the doubled grid
    selects -- with per-group weights and bias, shared or per-group inputs, both residuals and the relu second output."""
    G = 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    Co = Cout // (shuffle * shuffle) if shuffle else Cout
    xs = [split(nhwc(rnd(B, Cin, H, W, seed=1 + (0 if shared else g)))).to(DEV) for g in range(G)]
    ws = [split(rnd(Cout, k, k, Cin, seed=20 + g, scale=(Cin * k * k) ** -0.5)).to(DEV) for g in range(G)]
    bs = [rnd(Co, seed=30 + g, scale=0.1).to(DEV) for g in range(G)]
    res = [[split(rnd(B, Ho, Wo, Cout, seed=40 + 10 * i + g)).to(DEV) for g in range(G)] for i in range(nres)]
    oshape = (2, B, Ho * shuffle, Wo * shuffle, Co) if shuffle else (2, B, Ho, Wo, Cout)
    zero = torch.zeros(256, device=DEV)
    want, want_r = [], []
    for g in range(G):
        out = torch.full(oshape, 7.0, device=DEV, dtype=torch.bfloat16)
        orl = None if shuffle else torch.full(oshape, 7.0, device=DEV, dtype=torch.bfloat16)
        hip.conv2d_x3(xs[g], B, H, W, Cin, ws[g], Cout, k, k, stride, pad, out, zero, bias=bs[g], act=act,
                      res1=res[0][g] if nres > 0 else None, res2=res[1][g] if nres > 1 else None, shuffle=shuffle, out_relu=orl)
        want.append(out)
        want_r.append(orl)
    cat = lambda ts: torch.cat(ts, dim=1).contiguous()  # noqa: E731  (2, G*B, ...): planes outermost, groups stacked on the batch
    xg = xs[0] if shared else cat(xs)
    wg = torch.stack(ws, dim=1).contiguous()   # (2, G, Cout, k, k, Cin)
    bg = torch.stack(bs, dim=0).contiguous()   # (G, Co)
    gshape = (2, G * B) + tuple(oshape[2:])
    out = torch.full(gshape, 7.0, device=DEV, dtype=torch.bfloat16)
    orl = None if shuffle else torch.full(gshape, 7.0, device=DEV, dtype=torch.bfloat16)
    hip.conv2d_x3(xg, B, H, W, Cin, wg, Cout, k, k, stride, pad, out, zero, bias=bg, act=act,
                  res1=cat(res[0]) if nres > 0 else None, res2=cat(res[1]) if nres > 1 else None, shuffle=shuffle, out_relu=orl, groups=G, in_shared=shared)
    assert torch.equal(out.view(torch.int16), cat(want).view(torch.int16))
    if not shuffle:
        assert torch.equal(orl.view(torch.int16), cat(want_r).view(torch.int16))


@pytest.mark.parametrize(
    "H,W,Cin,Cout,k,stride,pad,nres",
    [
        (19, 19, 768, 256, 3, 1, 1, 0),   # 216 K-tiles -> 6 slices (layer_rn of the coarsest level)
        (37, 37, 256, 256, 3, 1, 1, 2),   # 72 K-tiles -> 3 slices, residuals + relu output (the 37^2 RCUs)
        (37, 37, 768, 768, 3, 2, 1, 0),   # stride 2 down to 19^2 (act_4_postprocess)
        (37, 37, 384, 256, 3, 1, 1, 0),   # 108 K-tiles -> 4 slices
        (21, 23, 96, 64, 3, 1, 1, 1),     # 27 K-tiles: below the threshold, the workspace must be left untouched
    ],
)
def test_conv2d_bf16x3_split_k_is_deterministic_and_batch_invariant(hip, H, W, Cin, Cout, k, stride, pad, nres):
    """Split-K of the small-map long-K layers (ufm_conv2d_nhwc_bf16x3_grouped with a workspace): (1) equals the unsplit kernel
    to fp32 re-association (2) repeated launches agree bit for bit (fixed range order, whoever arrives last) (3) an image's
    result is the same bits at batch 1 and inside a batch of 5 and inside a 2-group launch -- the split factor depends on the
    layer's geometry only (4) the tile counters are back at zero."""
    lib = hip.lib()
    B = 5
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x = split(nhwc(rnd(B, Cin, H, W, seed=1))).to(DEV)
    w = split(rnd(Cout, k, k, Cin, seed=2, scale=(Cin * k * k) ** -0.5)).to(DEV)
    b = rnd(Cout, seed=3, scale=0.1).to(DEV)
    res = [split(rnd(B, Ho, Wo, Cout, seed=10 + i)).to(DEV) for i in range(nres)] + [None, None]
    zero = torch.zeros(256, device=DEV)
    need = lib.ufm_conv_x3_splitk_ws_bytes(2, B, H, W, Cin, Cout, k, k, stride, pad)
    expect_split = Ho * Wo <= 1600 and k * k * (Cin // 32) >= 48
    assert (need > 0) == expect_split
    ws = torch.zeros(max(need, 1 << 16) // 4, device=DEV, dtype=torch.float32)

    def run(xx, bb, r, nb, groups=1, ww=w, use_ws=True):
        out = torch.full((2, groups * nb, Ho, Wo, Cout), 7.0, device=DEV, dtype=torch.bfloat16)
        orl = torch.full_like(out, 7.0)
        hip.conv2d_x3(xx, nb, H, W, Cin, ww, Cout, k, k, stride, pad, out, zero, bias=bb, res1=r[0], res2=r[1], out_relu=orl, groups=groups, splitk_ws=ws if use_ws else None)
        return out, orl

    plain, _ = run(x, b, res, B, use_ws=False)
    a, ar = run(x, b, res, B)
    assert float(ws[: 1 << 14].abs().max()) == 0.0  # counters (the first 64 KiB) are zero again
    if not expect_split:
        assert torch.equal(a, plain) and float(ws.abs().max()) == 0.0
        return
    err = (unsplit(a.cpu()) - unsplit(plain.cpu())).abs().max().item()
    # a different summation order, nothing more: a few units of the split format's own resolution (2^-17 relative)
    assert 0 < err <= 4e-5 * max(1.0, unsplit(plain.cpu()).abs().max().item()), err
    for _ in range(3):
        a2, ar2 = run(x, b, res, B)
        assert torch.equal(a2, a) and torch.equal(ar2, ar)
    for i in (0, 3):  # one image alone == the same image inside the batch
        one, _ = run(x[:, i : i + 1].contiguous(), b, [r[:, i : i + 1].contiguous() if r is not None else None for r in res], 1)
        assert torch.equal(one[:, 0], a[:, i])
    # two groups (the second with other weights): group 0 == the ungrouped launch
    w2 = split(rnd(Cout, k, k, Cin, seed=5, scale=(Cin * k * k) ** -0.5)).to(DEV)
    cat = lambda t: torch.cat([t, t], dim=1).contiguous() if t is not None else None  # noqa: E731
    g, _ = run(cat(x), torch.stack([b, b * 2]), [cat(r) for r in res], B, groups=2, ww=torch.stack([w, w2], dim=1).contiguous())
    assert torch.equal(g[:, :B], a)


@pytest.mark.parametrize("s,Cin,Co", [(4, 32, 32), (2, 64, 48)])
def test_conv_transpose_bf16x3(hip, s, Cin, Co):
    B, H, W = 2, 5, 7
    x = rnd(B, Cin, H, W, seed=1)
    wt = rnd(Cin, Co, s, s, seed=2, scale=Cin**-0.5)
    b = rnd(Co, seed=3, scale=0.1)
    ref = F.conv_transpose2d(x, wt, b, stride=s)
    wp = wt.permute(2, 3, 1, 0).reshape(s * s * Co, Cin).contiguous()
    out = torch.zeros(2, B, H * s, W * s, Co, device=DEV, dtype=torch.bfloat16)
    hip.conv2d_x3(split(nhwc(x)).to(DEV), B, H, W, Cin, split(wp).to(DEV), s * s * Co, 1, 1, 1, 0, out, torch.zeros(256, device=DEV), bias=b.to(DEV), shuffle=s)
    assert (unsplit(out.cpu()).permute(0, 3, 1, 2) - ref).abs().max().item() <= 4e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B,h,w,H,W,Ct,kinds", [(2, 21, 19, 37, 33, 2, [0, 0]), (1, 37, 37, 64, 64, 1, [1]), (3, 9, 12, 16, 21, 3, [0, 1, 0])])
def test_dpt_tail_fused_bit_identical(hip, B, h, w, H, W, Ct, kinds):
    """ufm_dpt_tail_fused == ufm_upsample_bilinear_nhwc -> ufm_conv2d_nhwc_bf16x3(ReLU) -> ufm_head_tail, bit for bit
    (ragged tiles, image borders = the conv's zero padding, flow and mask adaptors), and close to the fp64 statement."""
    x = rnd(B, 128, h, w, seed=1)
    w2 = rnd(32, 128, 3, 3, seed=2, scale=(128 * 9) ** -0.5)
    b2 = rnd(32, seed=3, scale=0.1)
    wt, bt = rnd(Ct, 32, seed=4, scale=0.3), rnd(Ct, seed=5, scale=0.1)
    a, d = [1.5, 1.0, 0.5][:Ct], [0.25, 0.0, -1.0][:Ct]
    xs = split(nhwc(x)).to(DEV)
    w2s = split(w2.permute(0, 2, 3, 1).contiguous()).to(DEV)
    zero = torch.zeros(256, device=DEV)
    up = torch.zeros(2, B, H, W, 128, device=DEV, dtype=torch.bfloat16)
    hip.upsample_bilinear(xs, B, h, w, 128, up, H, W)
    c2 = torch.zeros(2, B, H, W, 32, device=DEV, dtype=torch.bfloat16)
    hip.conv2d_x3(up, B, H, W, 128, w2s, 32, 3, 3, 1, 1, c2, zero, bias=b2.to(DEV), act=2)
    ref_out = torch.full((B, Ct, H, W), 7.0, device=DEV)
    ref_log = torch.full((B, Ct, H, W), 7.0, device=DEV)
    hip.head_tail(c2, B * H * W, H * W, 32, wt.to(DEV), bt.to(DEV), Ct, kinds, a, d, ref_out, ref_log)
    out = torch.full((B, Ct, H, W), 7.0, device=DEV)
    log = torch.full((B, Ct, H, W), 7.0, device=DEV)
    hip.dpt_tail_fused(xs, B, h, w, 128, w2s, b2.to(DEV), 32, H, W, wt.to(DEV), bt.to(DEV), Ct, kinds, a, d, out, log)
    assert torch.equal(out.view(torch.int32), ref_out.view(torch.int32))
    assert torch.equal(log.view(torch.int32), ref_log.view(torch.int32))
    y = F.conv2d(F.relu(F.conv2d(F.interpolate(x.double(), size=(H, W), mode="bilinear", align_corners=True), w2.double(), b2.double(), padding=1)),
                 wt.double()[:, :, None, None], bt.double())
    for c in range(Ct):
        want = torch.sigmoid(y[:, c]) if kinds[c] == 1 else y[:, c] * a[c] + d[c]
        assert (out.cpu()[:, c].double() - want).abs().max().item() <= 1e-4


@pytest.mark.parametrize("B,H,W,C,Ho,Wo,crop", [(2, 19, 23, 64, 38, 46, (0, 0)), (1, 37, 37, 256, 74, 74, (0, 0)), (2, 10, 13, 128, 20, 26, (19, 25)), (1, 40, 33, 64, 70, 58, (0, 0))])
def test_upsample_split_tiled_bit_identical(hip, B, H, W, C, Ho, Wo, crop):
    """The LDS-tiled split-format upsample (each source value loaded once per 8x32 tile) against the one-thread-per-output
    kernel: bit-identical, incl. ragged tiles, the cropped form after refinenet4 and the 296 -> 518-like ratio 0.57."""
    lib = hip.lib()
    x = split(rnd(B, H, W, C, seed=1)).to(DEV)
    Hs, Ws = crop[0] or Ho, crop[1] or Wo
    outs = []
    try:
        for tiled in (0, 1):
            lib.ufm_debug_set_upsample_variant(tiled)
            o = torch.full((2, B, Hs, Ws, C), 7.0, device=DEV, dtype=torch.bfloat16)
            hip.upsample_bilinear(x, B, H, W, C, o, Ho, Wo, crop[0], crop[1])
            outs.append(o.cpu())
    finally:
        lib.ufm_debug_set_upsample_variant(1)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    ref = F.interpolate(unsplit(x.cpu()).permute(0, 3, 1, 2), size=(Ho, Wo), mode="bilinear", align_corners=True)[:, :, :Hs, :Ws]
    assert (unsplit(outs[1]).permute(0, 3, 1, 2) - ref).abs().max().item() <= 6e-5 * max(1.0, ref.abs().max().item())


def test_split_format_layernorm_upsample_tail(hip):
    rows, D = 37, 128
    x = rnd(rows, D, seed=1, scale=3.0)
    w, b = 1 + rnd(D, seed=2, scale=0.1), rnd(D, seed=3, scale=0.1)
    ref = F.layer_norm(x, (D,), w, b, 1e-6)
    out = torch.zeros(2, rows, D, device=DEV, dtype=torch.bfloat16)
    hip.layernorm(x.to(DEV), D, None, rows, D, w.to(DEV), b.to(DEV), 1e-6, out, split=True)
    assert (unsplit(out.cpu()) - ref).abs().max().item() <= 4e-5  # 2^-17 relative of |y| <~ 4
    B, H, W, Cc = 2, 7, 9, 8
    xi = rnd(B, Cc, H, W, seed=4)
    refu = F.interpolate(xi, size=(14, 18), mode="bilinear", align_corners=True)[:, :, :13, :17]
    ou = torch.zeros(2, B, 13, 17, Cc, device=DEV, dtype=torch.bfloat16)
    hip.upsample_bilinear(split(nhwc(xi)).to(DEV), B, H, W, Cc, ou, 14, 18, 13, 17)
    assert (unsplit(ou.cpu()).permute(0, 3, 1, 2) - refu).abs().max().item() <= 6e-5
    HW, Cin = 35, 32
    xt, wt_, bt = rnd(B * HW, Cin, seed=1), rnd(2, Cin, seed=2), rnd(2, seed=3)
    y = (xt @ wt_.T + bt).reshape(B, HW, 2).permute(0, 2, 1)
    ot = torch.zeros(B, 2, HW, device=DEV)
    hip.head_tail(split(xt).to(DEV), B * HW, HW, Cin, wt_.to(DEV), bt.to(DEV), 2, [0, 1], [1.0, 1.0], [0.0, 0.0], ot, None)
    assert (ot.cpu()[:, 0] - y[:, 0]).abs().max() <= 2e-4 and (ot.cpu()[:, 1] - torch.sigmoid(y[:, 1])).abs().max() <= 1e-4


@pytest.mark.parametrize("split", [True, False])
def test_layernorm_slice_writes_only_its_rows_of_a_larger_buffer(hip, split):
    """ufm_layernorm_slice (engine: the micro-batch streams' pyramid levels land in one full-batch buffer): rows [r0, r0 + n)
    of a (2, R, D) split buffer -- lo plane R*D elements behind the hi plane -- or of a plain (R, D) one equal the stand-alone
    LayerNorm bit for bit and every other row keeps its fill value."""
    R, D, r0, n = 50, 256, 13, 21
    x = rnd(n, D, seed=1, scale=3.0).to(DEV)
    w, b = (1 + rnd(D, seed=2, scale=0.1)).to(DEV), rnd(D, seed=3, scale=0.1).to(DEV)
    if split:
        want = torch.zeros(2, n, D, device=DEV, dtype=torch.bfloat16)
        hip.layernorm(x, D, None, n, D, w, b, 1e-6, want, split=True)
        full = torch.full((2, R, D), 7.0, device=DEV, dtype=torch.bfloat16)
        dst = full.narrow(-2, r0, n)
        hip.layernorm(x, D, None, n, D, w, b, 1e-6, dst, split=True, out_plane=R * D)
        assert torch.equal(full[:, r0 : r0 + n].view(torch.int16), want.view(torch.int16))
        assert torch.all(full[:, :r0] == 7.0) and torch.all(full[:, r0 + n :] == 7.0)
    else:
        want = torch.zeros(n, D, device=DEV)
        hip.layernorm(x, D, None, n, D, w, b, 1e-6, want)
        full = torch.full((R, D), 7.0, device=DEV)
        hip.layernorm(x, D, None, n, D, w, b, 1e-6, full.narrow(0, r0, n), out_plane=R * D)
        assert torch.equal(full[r0 : r0 + n], want)
        assert torch.all(full[:r0] == 7.0) and torch.all(full[r0 + n :] == 7.0)


# ----------------------------------------------------------------------------- UNet layout kernels (R5)
def to_fmt(x_nhwc, split):
    """fp32 NHWC -> device buffer in the head format (fp32, or the (2, ...) bf16 split planes)."""
    if not split:
        return x_nhwc.to(DEV).contiguous()
    hi = x_nhwc.to(torch.bfloat16)
    lo = (x_nhwc - hi.float()).to(torch.bfloat16)
    return torch.stack([hi, lo]).to(DEV).contiguous()


def from_fmt(t):
    return (t[0].float() + t[1].float()).cpu() if t.dtype == torch.bfloat16 else t.cpu()


@pytest.mark.parametrize("split", [False, True])
def test_unet_layout_kernels(hip, split):
    """nn.MaxPool2d(2,2), F.interpolate(size) legacy nearest into a concat slot, image normalise + NHWC pad
    (unet_encoder.py:37-68; base.py:228-229) against torch, fp32 and split format, odd sizes."""
    B, H, W, C = 2, 13, 27, 16
    x = rnd(B, H, W, C, seed=1)
    if split:
        x = from_fmt(to_fmt(x, True))  # representable values: the kernels move (hi, lo) pairs, they never re-round
    xd = to_fmt(x, split)
    shp = lambda *s: ((2,) + s) if split else s  # noqa: E731
    dt = torch.bfloat16 if split else torch.float32
    out = torch.zeros(shp(B, H // 2, W // 2, C), device=DEV, dtype=dt)
    hip.maxpool2x2(xd, B, H, W, C, out)
    want = F.max_pool2d(x.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    assert torch.equal(from_fmt(out), want)
    # nearest resize 6x13 -> 13x27 into channels [8, 24) of a 32-channel buffer + a same-size copy into [0, 8)
    small = rnd(B, 6, 13, C, seed=2)
    if split:
        small = from_fmt(to_fmt(small, True))
    cat = torch.zeros(shp(B, H, W, 32), device=DEV, dtype=dt)
    hip.resize_nearest(to_fmt(small, split), B, 6, 13, C, cat, H, W, 32, 8)
    hip.resize_nearest(to_fmt(x[..., :8].contiguous(), split), B, H, W, 8, cat, H, W, 32, 0)
    got = from_fmt(cat)
    want = F.interpolate(small.permute(0, 3, 1, 2), size=(H, W)).permute(0, 2, 3, 1)
    assert torch.equal(got[..., 8:24], want) and torch.equal(got[..., :8], x[..., :8]) and torch.all(got[..., 24:] == 0)
    # image -> NHWC, 3 -> 32 channels, uint8 BHWC and float BCHW
    img = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(3))
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    o = torch.full(shp(B, H, W, 32), 7.0, device=DEV, dtype=dt)
    hip.image_to_nhwc(img.to(DEV), 0, B, H, W, std, mean, o, 32)
    want = (img.float() / 255.0 - torch.tensor(mean)) / torch.tensor(std)
    got = from_fmt(o)
    assert (got[..., :3] - want).abs().max() <= (1e-6 if not split else 2e-5) and torch.all(got[..., 3:] == 0)
    imgf = rnd(B, 3, H, W, seed=4)
    hip.image_to_nhwc(imgf.to(DEV), 1, B, H, W, [1.0] * 3, [0.0] * 3, o, 32)
    assert (from_fmt(o)[..., :3] - imgf.permute(0, 2, 3, 1)).abs().max() <= (0 if not split else 2e-5)


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("method", [0, 1])
def test_unet_combine(hip, split, method):
    """ufm.py:967-983: conv2(relu(conv1(cat[cls, unet]))) / conv2(cls * tanh(unet)) per pixel."""
    N, H, W = 2, 9, 11
    cls = rnd(N, 16, H, W, seed=1)
    un = rnd(N, H, W, 32, seed=2)
    if split:
        un = from_fmt(to_fmt(un, True))
    w1, b1 = rnd(32, 32, seed=3, scale=0.2), rnd(32, seed=4, scale=0.1)
    w2, b2 = rnd(16, 32 if method == 0 else 16, seed=5, scale=0.2), rnd(16, seed=6, scale=0.1)
    unc = un[..., :16].permute(0, 3, 1, 2)
    if method == 0:
        want = F.conv2d(F.relu(F.conv2d(torch.cat([cls, unc], 1), w1[:, :, None, None], b1)), w2[:, :, None, None], b2)
    else:
        want = F.conv2d(cls * torch.tanh(unc), w2[:, :, None, None], b2)
    out = torch.zeros(N, 16, H, W, device=DEV)
    hip.unet_combine(cls.to(DEV), to_fmt(un, split), N, H * W, 32, w1.to(DEV), b1.to(DEV), w2.to(DEV), b2.to(DEV), method, out)
    assert (out.cpu() - want).abs().max() <= 2e-5
