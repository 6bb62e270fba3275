#!/usr/bin/env python3
"""In-kernel stamps of the GEMM kernels (diagnostic instantiations, ufm_debug_set_gemm_stamps): the clock the chip holds under
the kernel (s_memtime / s_memrealtime after >= 2 s of back-to-back launches on random data -- MI355X_MICROARCH.md "DVFS
give-back" item 6) and the per-CU timeline of K loops and epilogues: how much of a CU's time has a K loop running on it, and how
much of the epilogue time of the pair kernel (two resident workgroups per CU) lies under the neighbour's K loop."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufm_amd import hip
lib = hip.lib()

SHAPES = {"proj": (21920, 1024, 1024, "res"), "fc2": (21920, 1024, 4096, "res"), "fc1": (21920, 4096, 1024, "gelu"),
          "i_proj": (21904, 768, 768, "res"), "i_fc2": (21904, 768, 3072, "res")}
WARM_S = float(os.environ.get("WARM_S", "2.0"))


def stamped(name, variant, rows=0, flags=0):
    M, N, K, mode = SHAPES[name]
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * K**-0.5).bfloat16()
    bias = torch.randn(N, device="cuda") * 0.1
    gamma = 1 + 0.1 * torch.randn(N, device="cuda")
    out = torch.randn(M, N, device="cuda") if mode == "res" else torch.empty(M, N, device="cuda", dtype=torch.bfloat16)

    def run():
        hip.gemm_bf16(A, W, M, N, K, out, bias=bias, act=hip.ACT_GELU if mode == "gelu" else hip.ACT_NONE, res=out if mode == "res" else None,
                      gamma=gamma if mode == "res" else None)
    lib.ufm_debug_set_gemm_variant(variant); lib.ufm_debug_set_gemm_tile_rows(rows); lib.ufm_debug_set_gemm_flags(flags)
    nrows = 8192
    buf = torch.zeros(nrows * 8, device="cuda", dtype=torch.int64)
    t0 = time.time()
    while time.time() - t0 < WARM_S:  # the clock settles under load
        for _ in range(50): run()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    plain_us = e0.elapsed_time(e1) / 20 * 1e3
    hip._check(lib.ufm_debug_set_gemm_stamps(buf.data_ptr(), nrows), "ufm_debug_set_gemm_stamps")
    for _ in range(20): run()
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    stamped_us = e0.elapsed_time(e1) / 20 * 1e3
    hip._check(lib.ufm_debug_set_gemm_stamps(None, 0), "ufm_debug_set_gemm_stamps")
    lib.ufm_debug_set_gemm_variant(0); lib.ufm_debug_set_gemm_tile_rows(0); lib.ufm_debug_set_gemm_flags(0)
    d = buf.view(nrows, 8).cpu()
    d = d[d[:, 4] != 0]
    return d, plain_us, stamped_us, 2.0 * M * N * K


def analyse(d):
    import numpy as np
    a = d.numpy().astype(np.uint64)
    hw = (a[:, 0] >> np.uint64(32)).astype(np.int64); xcc = (a[:, 1] & np.uint64(0xf)).astype(np.int64)
    lds_base = ((a[:, 1] >> np.uint64(32)) & np.uint64(0x1ff)).astype(np.int64)
    t_in, t_loop, t_end, rt_in, rt_end, t_pro = (a[:, i].astype(np.float64) for i in (2, 3, 4, 5, 6, 7))
    clock = np.median((t_end - t_in) / np.maximum(rt_end - rt_in, 1) * 0.1)
    cu = (xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
    res = {"workgroups": int(len(a)), "cus": int(len(set(cu.tolist()))), "clock_ghz": float(clock),
           "prologue_cycles": float(np.median(t_pro - t_in)), "loop_cycles": float(np.median(t_loop - t_pro)), "epilogue_cycles": float(np.median(t_end - t_loop)),
           "second_residents": int((lds_base != 0).sum())}
    # per-CU timeline (s_memtime is one counter per XCD at least; compare only within a CU): union of K-loop intervals, epilogue time under a K loop
    span = loop_cover = epi = epi_under = 0.0
    for c in set(cu.tolist()):
        m = cu == c
        lo, hi = t_in[m].min(), t_end[m].max()
        span += hi - lo
        iv = sorted(zip(t_in[m], t_loop[m]))
        cur_s, cur_e, cov = iv[0][0], iv[0][1], 0.0
        for s, e in iv[1:]:
            if s > cur_e: cov += cur_e - cur_s; cur_s, cur_e = s, e
            else: cur_e = max(cur_e, e)
        cov += cur_e - cur_s
        loop_cover += cov
        for s, e in zip(t_loop[m], t_end[m]):
            epi += e - s
            u = 0.0
            for ls, le in zip(t_in[m], t_loop[m]):
                u += max(0.0, min(e, le) - max(s, ls))
            epi_under += min(u, e - s)
    if os.environ.get("DUMP"):  # the timelines of a few CUs, cycles from the XCD's first entry: (entry, K loop from, to, end, second resident)
        for c in sorted(set(cu.tolist()))[:: max(1, len(set(cu.tolist())) // int(os.environ["DUMP"]))][: int(os.environ["DUMP"])]:
            m = cu == c
            base = t_in[xcc == (c >> 12)].min()
            rows = sorted(zip(t_in[m] - base, t_pro[m] - base, t_loop[m] - base, t_end[m] - base, lds_base[m] != 0))
            print(f"  cu 0x{c:04x}: " + "  ".join(f"[{a/1e3:.1f} {b/1e3:.1f} {cc/1e3:.1f} {d/1e3:.1f}{' *' if e else ''}]" for a, b, cc, d, e in rows), flush=True)
        for x in sorted(set(xcc.tolist())):
            mx = xcc == x
            print(f"  xcd {x}: wall {(t_end[mx].max() - t_in[mx].min())/1e3:.1f} k cycles, last entry at {(t_in[mx].max() - t_in[mx].min())/1e3:.1f}", flush=True)
    res["cu_time_with_a_k_loop_running"] = loop_cover / span
    res["epilogue_time_under_a_neighbours_k_loop"] = epi_under / max(epi, 1.0)
    res["kernel_cycles_per_cu"] = span / res["cus"]
    return res


if __name__ == "__main__":
    names = os.environ.get("SHAPES", "proj,fc2,fc1").split(",")
    arms = [("8-phase 192 rows", 4, 192, 0), ("8-phase 256 rows", 4, 256, 0), ("pair", 6, 0, 0), ("pair stagger 3", 6, 0, 3 << 16)]
    if os.environ.get("ARMS"):
        arms = [(f"pair stagger {int(x)}", 6, 0, int(x) << 16) if x.isdigit() else (f"8-phase {x[1:]} rows", 4, int(x[1:]), 0) for x in os.environ["ARMS"].split(",")]
    out = {}
    for n in names:
        for label, v, rows, flags in arms:
            d, plain, st, fl = stamped(n, v, rows, flags)
            r = analyse(d)
            r.update(plain_us=plain, stamped_us=st, tflops=fl / plain / 1e6, frac=fl / plain / 1e6 / 2500.0)
            r["frac_at_clock"] = r["tflops"] / (2500.0 * r["clock_ghz"] / 2.4)
            out[f"{n} | {label}"] = r
            print(f"{n:7s} {label:18s}", json.dumps({k: (round(x, 4) if isinstance(x, float) else x) for k, x in r.items()}), flush=True)
    print(json.dumps(out))
