#!/usr/bin/env python3
"""Headline benchmark: image-pairs/sec + p50 latency, UFM-Base 518x518, N x MI355X (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without WORLD_SIZE: starts the line below itself, as a child process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one ``predict_correspondences_batched`` over ``--batch`` (default 8) synthetic
518x518 uint8 pairs per GPU (BASELINE config 2: "UFM-Base, random-init weights, batch=8
518x518 synthetic pairs"), inputs resident in HBM before the timed region.  N > 1 = batch-split
data parallel (independent pairs, weak scaling, no data-path collective; a single RCCL all_gather
of the packed [flow|covisibility] results closes every step).

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      dominant MFMA kernel family (by measured time) of the step: algorithmic FLOPs
                per launch / average launch duration (HIP events on the launch stream), vs the
                dense MFMA peak for its operand dtype; plus `attention` (the north star's
                "attention-GEMM roofline") and per-family breakdown under `kernels`.
  cpu_baseline  the fp32 CPU oracle (a port; the reference's own CPU path cannot run, its
                arithmetic lives in an absent package) timed on this box's host cores on a
                bounded sample (rank 0, N=1 only).
  precise_mode  the SAME workload in numerics "precise" (bf16x3 split precision for every contraction: the 1e-3 px gate on
                the bf16 matrix cores): pairs/s, ms/step, its flow max-abs vs the oracle, per-kernel durations (rank 0, N=1).
  parity_mode   the SAME workload in numerics "parity" (exact-fp32 MFMA everywhere, the mode that meets the
                1e-3 px gate): pairs/s, ms/step and its flow max-abs vs the oracle (rank 0, N=1 only).
  latency_b1_ms single-pair p50 latency (BASELINE metric "pairs/s + p50 latency"): eager launches and
                HIP-graph replay (ufm_amd.GraphedPredictor).
  config4 / config5   BASELINE.json's other single-GPU configurations under the same clock (side legs, rank 0, N=1): UFM-Refine
                518^2 batch 8 and UFM-Base 1036^2 batch 2 -- pairs/s, ms/step, per-family MFMA fractions (attention at N = 10 954).
  pipeline_kernels   the per-family table of the dispatch the headline runs: instrumented TWO-stream steps (events per launch on each micro-batch /
                head stream); `roofline` stays the single-stream leg (kernels do not overlap there) and says which tile policy each was measured under.
  default_res   the reference's class-default resolution (inference_resolution=None -> 560 x 420, base.py:89-90) at B = 8 on 1080 x 810 inputs (side leg).
  roofline.clock_ghz / frac_at_clock   the clock the chip holds under the dominant family (s_memtime / s_memrealtime stamps of the
                diagnostic GEMM instantiations after 2 s of load) and `achieved` against the peak AT that clock.
"""

import argparse
import glob
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip table)
PEAK_F32_TFLOPS = 157.3    # f32-input MFMA
PEAK_BF16X3_TFLOPS = PEAK_BF16_TFLOPS / 3.0  # split-precision conv: 3 bf16 MFMAs per algorithmic MAC
PEAK_HBM_GBS = 8000.0


# MFMA-bound kernel families and the dense peak their operand format allows
MFMA_PEAKS = {"ufm_gemm_bf16": PEAK_BF16_TFLOPS, "ufm_attention_bf16": PEAK_BF16_TFLOPS, "ufm_conv2d_nhwc_f32": PEAK_F32_TFLOPS,
              "ufm_attention_f32": PEAK_F32_TFLOPS, "ufm_conv2d_nhwc_bf16x3": PEAK_BF16X3_TFLOPS, "ufm_dpt_tail_fused": PEAK_BF16X3_TFLOPS,
              "ufm_gemm_bf16x3": PEAK_BF16X3_TFLOPS, "ufm_attention_bf16x3": PEAK_BF16X3_TFLOPS}


def meta_work(m) -> float:
    """A launch's algorithmic work: KernelTimer metas are a number or (number, shape tag)."""
    if not m:
        return 0.0
    return float(m[0]) if isinstance(m, tuple) else float(m)


class _Ms:
    """A pair of stand-in 'events' whose elapsed_time is a precomputed duration (ms): lets the median-of-steps records below go
    through the same tables as raw (name, event, event, meta) records."""

    def __init__(self, ms: float):
        self.ms = ms

    def elapsed_time(self, other) -> float:
        return other.ms - self.ms


def instrumented_steps(run, reps: int = 3, concurrent: bool = False):
    """`reps` instrumented steps (HIP events around every C-ABI launch, on the launch stream of the launching thread); per launch -- the steps
    issue the same launches in the same order ON EACH STREAM -- the MEDIAN duration over the steps.  One step alone carries first-touch outliers (the
    single-stream workspace is first used here: one QKV launch of 24 at 600 us moved that shape's average from 133 to 160 us in a round-5 run).
    Works for the engine's two-stream steps too: launches are keyed by (stream, position on that stream).
    Returns (summary, records) in hip.KernelTimer's formats."""
    from ufm_amd import hip as _hip
    import torch as _torch

    per_step = []
    for _ in range(reps):
        _hip.TIMER = _hip.KernelTimer(concurrent=concurrent)
        try:
            run()
            _torch.cuda.synchronize()
            recs, strs = list(_hip.TIMER.records), list(_hip.TIMER.streams)
        finally:
            _hip.TIMER = None
        by_stream = {}
        for (name, e0, e1, meta), st in zip(recs, strs):
            by_stream.setdefault(st, []).append((name, e0.elapsed_time(e1), meta))
        per_step.append([r for st in sorted(by_stream) for r in by_stream[st]])
    base = per_step[-1]
    same = [st for st in per_step if len(st) == len(base) and all(a[0] == b[0] for a, b in zip(st, base))]
    records, summ = [], {}
    for i, (name, _ms, meta) in enumerate(base):
        v = sorted(st[i][1] for st in same)
        ms = v[len(v) // 2]
        records.append((name, _Ms(0.0), _Ms(ms), meta))
        d = summ.setdefault(name, dict(ms=0.0, launches=0, metas=[]))
        d["ms"] += ms
        d["launches"] += 1
        d["metas"].append(meta)
    return summ, records


def pipeline_families(model, src, tgt, wall_ms: float) -> dict:
    """the per-family table of the dispatch the HEADLINE runs -- three instrumented TWO-stream steps (the timed configuration:
    micro-batch streams flagged with ufm_hint_concurrent_stream, CU-time tile policy, the heads on their own streams), HIP events around every
    launch on its own stream, per-launch medians.  Launches of different streams overlap in time, so a family's summed launch time is NOT its
    share of the step: `frac` here = algorithmic work / summed launch durations / peak is the rate a launch sustains WHILE SHARING the chip with
    the other stream's kernels, and `sum_ms / wall` says how much of it overlapped."""
    summ, records = instrumented_steps(lambda: model.predict_correspondences_batched(src, tgt), concurrent=True)
    fam, total = {}, 0.0
    for name, d in summ.items():
        total += d["ms"]
        e = {"launches": d["launches"], "sum_ms": round(d["ms"], 3)}
        if name in MFMA_PEAKS:
            work = sum(meta_work(m) for m in d["metas"])
            e["frac_while_sharing"] = round(work / (d["ms"] * 1e-3) / 1e12 / MFMA_PEAKS[name], 4)
            if name == "ufm_gemm_bf16":
                e["per_shape_frac"] = {t.replace(" (read-modify-write)", "").replace(" out", ""): r["frac"] for t, r in per_shape_table([r for r in records if r[0] == name], MFMA_PEAKS[name]).items() if r["launches"] > 1}
        fam[name.replace("ufm_", "")] = e
    return {"families": fam, "sum_of_launch_ms": round(total, 3), "wall_ms_per_step": round(wall_ms, 3), "overlap_factor": round(total / wall_ms, 3),
            "dispatch": "two micro-batch streams flagged concurrent (CU-time tile policy), heads on two streams: the timed configuration",
            "how": "three instrumented two-stream steps, HIP events around every launch on its own stream, per-launch medians keyed by (stream, position)"}


def per_shape_table(d, peak_tflops: float):
    """GEMM launches of one instrumented step grouped by shape tag (QKV / proj / fc1 / fc2 of the encoder and of the
    info-sharing blocks differ in M, N, K and epilogue): launches, total and average time, TFLOP/s, fraction of peak."""
    rows = {}
    for (name, e0, e1, meta) in d:
        if not isinstance(meta, tuple):
            continue
        r = rows.setdefault(meta[1], {"launches": 0, "ms": 0.0, "gflop": 0.0})
        r["launches"] += 1
        r["ms"] += e0.elapsed_time(e1)
        r["gflop"] += meta[0] / 1e9
    out = {}
    for tag, r in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]):
        tf = r["gflop"] / r["ms"] if r["ms"] > 0 else 0.0  # GFLOP / ms = TFLOP/s
        out[tag] = {"launches": r["launches"], "ms_per_step": round(r["ms"], 4), "avg_launch_us": round(1e3 * r["ms"] / r["launches"], 2),
                    "tflops": round(tf, 1), "frac": round(tf / peak_tflops, 4)}
    return out


def order_line(line: dict, pairs_per_step: int) -> dict:
    """The ONE JSON line, ordered for a reader who keeps only its tail: the bulky objects (per-kernel tables, config and
    sample descriptions) first, the contract's scalars next, and a compact `summary` of every judged number LAST.  Adds the
    two roofline figures SURVEY 8(d) asks for besides the per-kernel ones: `end_to_end` (all algorithmic MFMA FLOPs of a
    step x steps/s vs the bf16 peak) and `attention_share` (the attention-GEMM FLOPs of a step x steps/s vs the peak: the
    north star's "fraction of the attention-GEMM roofline" in its end-to-end form)."""
    out = {}
    kernels = line.get("kernels")
    pm = dict(line["precise_mode"]) if "precise_mode" in line else None
    if kernels is not None:
        out["kernels"] = kernels
    if pm is not None and "kernels" in pm:
        out["precise_mode_kernels"] = pm.pop("kernels")
    for k in ("config", "cpu_baseline"):
        if k in line:
            out[k] = line[k]
    summary = {}
    if kernels is not None and line.get("n_gpus") == 1:
        steps_per_s = line["value"] / pairs_per_step
        mfma = {k: v for k, v in kernels.items() if v.get("bound") == "mfma"}
        gf = sum(v["algorithmic_gflop"] for v in mfma.values())
        line["end_to_end"] = {"algorithmic_gflop_per_step": gf, "tflops": gf * steps_per_s / 1e3, "peak": PEAK_BF16_TFLOPS, "frac": gf * steps_per_s / 1e3 / PEAK_BF16_TFLOPS}
        att = next((v for k, v in mfma.items() if k.startswith("ufm_attention")), None)
        if att is not None:
            line["attention_share"] = {"attention_gflop_per_pair": att["algorithmic_gflop"] / pairs_per_step, "tflops": att["algorithmic_gflop"] * steps_per_s / 1e3,
                                       "frac": att["algorithmic_gflop"] * steps_per_s / 1e3 / PEAK_BF16_TFLOPS}
        summary["family_frac"] = {k.replace("ufm_", ""): round(v["frac"], 4) for k, v in kernels.items() if "frac" in v}
        summary["family_ms"] = {k.replace("ufm_", ""): round(v["ms_per_step"], 3) for k, v in kernels.items()}
        ps = kernels.get("ufm_gemm_bf16", {}).get("per_shape")
        if ps:
            summary["gemm_shape_frac"] = {t.replace(" (read-modify-write)", "").replace(" out", ""): r["frac"] for t, r in ps.items() if r["launches"] > 1}
    for k, v in line.items():
        if k not in out and k not in ("kernels", "precise_mode", "roofline", "attention", "check_vs_oracle", "latency_b1_ms", "parity_mode", "end_to_end", "attention_share", "config4", "config5", "default_res", "pipeline_kernels"):
            out[k] = v
    if pm is not None:
        out["precise_mode"] = pm
    if "pipeline_kernels" in line:
        out["pipeline_kernels"] = line["pipeline_kernels"]
    for k in ("config4", "config5", "default_res", "parity_mode", "check_vs_oracle", "latency_b1_ms", "attention", "end_to_end", "attention_share", "roofline"):
        if k in line:
            out[k] = line[k]
    # compact repeat of the judged scalars, last
    summary.update({"value": round(line["value"], 2), "ms_per_step": round(line["ms_per_step"], 3), "n_gpus": line["n_gpus"]})
    if "roofline" in line:
        summary["roofline_frac"] = round(line["roofline"]["frac"], 4)
    if "attention" in line:
        summary["attention_frac"] = round(line["attention"]["frac"], 4)
    if "end_to_end" in line:
        summary["end_to_end_frac"] = round(line["end_to_end"]["frac"], 4)
        summary["attention_share_frac"] = round(line["attention_share"]["frac"], 4) if "attention_share" in line else None
    if pm is not None:
        summary["precise"] = {"pairs_per_s": round(pm["value"], 2), "flow_max_abs": pm.get("flow_max_abs")}
    if "parity_mode" in line:
        summary["parity"] = {"pairs_per_s": round(line["parity_mode"]["value"], 2), "flow_max_abs": line["parity_mode"].get("flow_max_abs")}
    if "check_vs_oracle" in line:
        summary["fast_flow_max_abs"] = line["check_vs_oracle"]["flow_max_abs"]
    if "latency_b1_ms" in line:
        summary["latency_b1_ms"] = {k: round(v, 3) for k, v in line["latency_b1_ms"].items() if k.endswith("p50")}
    if "cpu_baseline" in line:
        summary["cpu_pairs_per_s"] = round(line["cpu_baseline"]["value"], 4)
    if "roofline" in line and "clock_ghz" in line["roofline"]:
        summary["clock_ghz"] = round(line["roofline"]["clock_ghz"], 3)
        summary["roofline_frac_at_clock"] = round(line["roofline"]["frac_at_clock"], 4)
        cv = (kernels or {}).get("ufm_conv2d_nhwc_bf16x3", {})
        if "clock_ghz" in cv:
            summary["conv_clock_ghz"], summary["conv_frac_at_clock"] = round(cv["clock_ghz"], 3), round(cv["frac_at_clock"], 4)
    if "pipeline_kernels" in line:
        pk = line["pipeline_kernels"]
        summary["pipeline_frac_while_sharing"] = {k: v["frac_while_sharing"] for k, v in pk["families"].items() if "frac_while_sharing" in v}
        summary["pipeline_overlap_factor"] = pk["overlap_factor"]
    for k in ("config4", "config5", "default_res"):
        if k in line:
            c = line[k]
            summary[k] = {"pairs_per_s": round(c["value"], 2), "ms_per_step": round(c["ms_per_step"], 2),
                          "attention_frac": next((v["frac"] for n, v in c["mfma_families"].items() if n.startswith("attention")), None)}
    for k in ("gather_check", "per_rank_ms_per_step", "gather_wait_ms"):
        if k in line:
            summary[k] = line[k]
    out["summary"] = summary
    return out


def host_cores() -> int:
    """Threads the CPU baseline may use: the cgroup CPU quota if one is set, else the affinity mask,
    capped at 16 (a 1-GPU box's CPU share; os.cpu_count() reports the whole 256-thread host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def csrc_sha256() -> str:
    """Content hash of ufm_amd/csrc (same function as tools/pmc_summarize.py): ties the PMC traffic summary to the
    kernel sources it was measured on."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "ufm_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def side_config(ufm_amd, hip, which: str, res: int, batch: int, steps: int, micro_batches: int) -> dict:
    """One of BASELINE.json's other single-GPU configurations, timed like the headline step (inputs resident, synthetic uint8 pairs,
    random-init weights): config 4 = UFM-Refine (UniFlowMatchClassificationRefinement, /root/reference/uniflowmatch/models/ufm.py:843-1009)
    at 518^2, batch 8; config 5 = UFM-Base at 1036^2 (5477-token encoder / 10 954-token joint attention), batch 2."""
    from ufm_amd.modules import init_weights_

    in_hw = (res, res)
    if which == "refine":
        m = ufm_amd.UniFlowMatchClassificationRefinement(**ufm_amd.ufm_refine_config(resolution_wh=(res, res))).eval()
    elif which == "default_res":
        # the reference's class default: inference_resolution=None -> (560, 420) W x H (/root/reference/uniflowmatch/models/base.py:89-90), a
        # 30 x 40 patch grid with the 518-native position embedding interpolated; inputs 1080 x 810 uint8 (antialiased resize on the GPU inside the step)
        cfg = ufm_amd.ufm_base_config()
        cfg.pop("inference_resolution")
        m = ufm_amd.UniFlowMatchConfidence(**cfg).eval()
        assert m.inference_resolution == [(560, 420)]
        in_hw = (810, 1080)
    else:
        m = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_base_config(resolution_wh=(res, res))).eval()
    init_weights_(m, seed=0)
    m = m.to("cuda").set_numerics("fast")
    m.engine().micro_batches = micro_batches
    g = torch.Generator().manual_seed(4321)
    s = torch.randint(0, 256, (batch, in_hw[0], in_hw[1], 3), dtype=torch.uint8, generator=g).cuda()
    t = torch.randint(0, 256, (batch, in_hw[0], in_hw[1], 3), dtype=torch.uint8, generator=g).cuda()
    for _ in range(2):
        m.predict_correspondences_batched(s, t)
    torch.cuda.synchronize()
    c0 = time.perf_counter()
    for _ in range(steps):
        m.predict_correspondences_batched(s, t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - c0) / steps
    what = (f"class-default inference resolution 560x420 (30 x 40 patch grid), {in_hw[1]}x{in_hw[0]} uint8 synthetic pairs resized on the GPU" if which == "default_res"
            else f"{res}x{res} synthetic pairs")
    out = {"workload": ("UFM-Refine" if which == "refine" else "UFM-Base") + f", random init, batch={batch} {what}, 1xMI355X",
           "value": batch / dt, "unit": "pairs/s", "ms_per_step": 1e3 * dt, "steps": steps, "numerics": "fast"}
    summ, _records = instrumented_steps(lambda: m.predict_correspondences_batched(s, t), reps=2)
    fam = {}
    for name, d in summ.items():
        if name in MFMA_PEAKS:
            work = sum(meta_work(x) for x in d["metas"])
            fam[name.replace("ufm_", "")] = {"ms": round(d["ms"], 3), "frac": round(work / (d["ms"] * 1e-3) / 1e12 / MFMA_PEAKS[name], 4)}
    out["mfma_families"] = fam
    del m
    torch.cuda.empty_cache()
    return out


def gemm_clock_under_load(hip) -> dict:
    """MI355X_MICROARCH.md, "DVFS give-back" item 6: the clock the chip holds under the dominant kernel family = delta s_memtime /
    delta s_memrealtime (100 MHz) stamped around whole workgroups of the diagnostic GEMM instantiations (the shipped kernels execute no
    stamp; the stamps go to a buffer no kernel reads), after >= 2 s of back-to-back launches of the encoder block's four GEMMs on random data."""
    M = 21920
    shapes = ((3072, 1024, "bf16"), (1024, 1024, "res"), (4096, 1024, "gelu"), (1024, 4096, "res"))
    ops = []
    for N, K, mode in shapes:
        A = torch.randn(M, K, device="cuda").bfloat16()
        W = (torch.randn(N, K, device="cuda") * K**-0.5).bfloat16()
        bias, gamma = torch.randn(N, device="cuda") * 0.1, 1 + 0.1 * torch.randn(N, device="cuda")
        out = torch.randn(M, N, device="cuda") if mode == "res" else torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ops.append((A, W, N, K, out, bias, gamma, mode))

    def block():
        for A, W, N, K, out, bias, gamma, mode in ops:
            hip.gemm_bf16(A, W, M, N, K, out, bias=bias, act=hip.ACT_GELU if mode == "gelu" else hip.ACT_NONE, res=out if mode == "res" else None,
                          gamma=gamma if mode != "gelu" else None)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 2.0:
        for _ in range(20):
            block()
        torch.cuda.synchronize()
    rows = 4096
    buf = torch.zeros(rows * 8, device="cuda", dtype=torch.int64)
    hip._check(hip.lib().ufm_debug_set_gemm_stamps(buf.data_ptr(), rows), "ufm_debug_set_gemm_stamps")
    try:
        for _ in range(10):
            block()
        torch.cuda.synchronize()
    finally:
        hip._check(hip.lib().ufm_debug_set_gemm_stamps(None, 0), "ufm_debug_set_gemm_stamps")
    d = buf.view(rows, 8).cpu()
    d = d[d[:, 4] != 0].double()
    clock = float(((d[:, 4] - d[:, 2]) / (d[:, 6] - d[:, 5]).clamp_min(1.0) * 0.1).median())
    del ops
    # the same for the heads' dominant layer, the 148^2 256 -> 256 3x3 residual convolution on the 8-phase bf16x3 kernel
    Bc, Hc, Cc = 8, 148, 256
    x = torch.randn(2, Bc, Hc, Hc, Cc, device="cuda").bfloat16()
    x[1] *= 2.0 ** -9
    w = (torch.randn(2, Cc, 3, 3, Cc, device="cuda") * (9 * Cc) ** -0.5).bfloat16()
    w[1] *= 2.0 ** -9
    bias, res1 = torch.randn(Cc, device="cuda") * 0.1, torch.randn(2, Bc, Hc, Hc, Cc, device="cuda").bfloat16()
    out, zero = torch.empty(2, Bc, Hc, Hc, Cc, device="cuda", dtype=torch.bfloat16), torch.zeros(256, device="cuda")

    def conv():
        hip.conv2d_x3(x, Bc, Hc, Hc, Cc, w, Cc, 3, 3, 1, 1, out, zero, bias=bias, res1=res1)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 2.0:
        for _ in range(20):
            conv()
        torch.cuda.synchronize()
    buf.zero_()
    hip._check(hip.lib().ufm_debug_set_conv_stamps(buf.data_ptr(), rows), "ufm_debug_set_conv_stamps")
    try:
        for _ in range(10):
            conv()
        torch.cuda.synchronize()
    finally:
        hip._check(hip.lib().ufm_debug_set_conv_stamps(None, 0), "ufm_debug_set_conv_stamps")
    dc = buf.view(rows, 8).cpu()
    dc = dc[dc[:, 4] != 0].double()
    conv_clock = float(((dc[:, 4] - dc[:, 2]) / (dc[:, 6] - dc[:, 5]).clamp_min(1.0) * 0.1).median()) if dc.shape[0] else None
    return {"clock_ghz": clock, "workgroups": int(d.shape[0]), "conv_clock_ghz": conv_clock,
            "how": "median over workgroups of delta s_memtime / delta s_memrealtime in the stamped fc1 / proj / fc2 instantiations, after 2 s of the encoder block's four GEMMs back to back (random data); "
                   "conv_clock_ghz: the same stamps in the 8-phase bf16x3 convolution kernel after 2 s of the 148^2 256->256 3x3 residual layer"}


def p50_ms(fn, iters: int, warm: int) -> float:
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    return sorted(ts)[len(ts) // 2]


def launch_command(n_gpus: int, argv, port: int):
    """The driver's own N > 1 command line: one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n_gpus: int, argv) -> int:
    """`python bench.py --gpus N` (N > 1) started without a launcher: run the ranks as a child `torch.distributed.run`, relay
    rank 0's single JSON line (stdout) and the ranks' stderr, return the child's exit code.  The parent makes no HIP call at all
    (a process that has initialised the GPU must neither exec nor fork workers on this pool): it only imports torch."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "2")  # the launcher would set 1; every rank runs two launch threads
    cmd = launch_command(n_gpus, argv, port)
    print(f"[bench launcher] --gpus {n_gpus} without WORLD_SIZE: starting {' '.join(cmd[1:8])} ... as a child process", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, cwd=ROOT)
    lines = 0
    for ln in proc.stdout:  # relay as it comes: exactly what the ranks print (rank 0's ONE JSON line)
        sys.stdout.write(ln)
        sys.stdout.flush()
        lines += ln.startswith("{")
    rc = proc.wait()
    print(f"[bench launcher] child exit code {rc}, JSON lines relayed {lines}, parent touched the GPU: {torch.cuda.is_initialized()}",
          file=sys.stderr, flush=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="pairs per GPU per step")
    ap.add_argument("--res", type=int, default=518)
    ap.add_argument("--numerics", default="fast", choices=["fast", "precise", "parity"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the extra numerics='parity' timing of the same workload")
    ap.add_argument("--no-precise-mode", action="store_true", help="skip the extra numerics='precise' timing of the same workload")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-pair latency measurement")
    ap.add_argument("--no-side-configs", action="store_true", help="skip the BASELINE config 4 (UFM-Refine 518^2) and config 5 (UFM-Base 1036^2) side legs")
    ap.add_argument("--no-clock", action="store_true", help="skip the in-kernel clock measurement of the dominant kernel family")
    ap.add_argument("--gemm-variant", type=int, default=0, help="tuning hook: ufm_debug_set_gemm_variant (0 = auto)")
    ap.add_argument("--attn-variant", type=int, default=0, help="tuning hook: ufm_debug_set_attn_variant (0 = default)")
    ap.add_argument("--concurrent-heads", type=int, default=-1, help="DPT heads on separate streams: -1 = engine default (automatic: only in single-stream forwards), 0 / 1 = force")
    ap.add_argument("--info-sharing", default="global_attention", choices=["global_attention", "cross_attention"],
                    help="side measurement (SURVEY 8(f)4): the cross-attention info-sharing variant (ufm.py:193) at UFM-Base dimensions")
    ap.add_argument("--rope", type=float, default=0.0, help="with --info-sharing cross_attention: RoPE-2D base frequency (0 = none)")
    ap.add_argument("--head", default="dpt", choices=["dpt", "moge_conv"], help="side measurement (SURVEY 8(f)4): the MoGe convolutional flow head (ufm.py:266-267)")
    ap.add_argument("--group-heads", type=int, default=0, help="1: the two DPT heads as one grouped launch per layer (Engine.group_heads; measured 1 %% slower than the default two launch sequences on two streams)")
    ap.add_argument("--conv-splitk", type=int, default=0, help="1: deterministic split-K in the heads' small-map convolutions (Engine.conv_splitk; measured neutral)")
    ap.add_argument("--last-layer-view1", type=int, default=1, help="0: the last joint-attention block on all rows (A/B of Engine.last_layer_view1)")
    ap.add_argument("--micro-batches", type=int, default=2, help="concurrent micro-batches (HIP streams) per GPU; 1 = single stream")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the ranks ourselves (as a CHILD process -- this parent never
        # touches the GPU) and leave with the launcher's return code
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # UFM_BENCH_SHARE_GPU=1 + UFM_BENCH_BACKEND=gloo: a FUNCTIONAL rehearsal of N > 1 on a one-GPU box (all ranks on cuda:0, the
    # device buffers gathered through gloo; RCCL refuses two ranks on one device).  Never a measurement: the line says so.
    share_gpu = os.environ.get("UFM_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # UFM_BENCH_FORCE_DIST=1: run the RCCL gather path with a single rank too (rehearsal of the N > 1 code on one GPU)
    use_dist = world > 1 or os.environ.get("UFM_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist_

        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:  # UFM_BENCH_FORCE_DIST=1 without a launcher: a one-rank group on the real backend
            for k_, v_ in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_PORT", "29533")):
                os.environ.setdefault(k_, v_)
        if os.environ.get("UFM_BENCH_BACKEND") == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
    # Host threads: every rank runs 2 Python launch threads (the two micro-batch streams).  With N ranks on one node the
    # intra-op pools of torch / OpenMP must not multiply on top of that (8 ranks x a 256-thread pool); only rank 0 at N = 1
    # runs the CPU-oracle leg, which sizes its own pool (host_cores()).
    host_threads = 2 if world > 1 else None
    if host_threads:
        torch.set_num_threads(host_threads)

    import ufm_amd
    from ufm_amd import hip
    from ufm_amd.modules import init_weights_

    hip.lib().ufm_debug_set_gemm_variant(args.gemm_variant)
    hip.lib().ufm_debug_set_attn_variant(args.attn_variant)
    res = args.res
    cfg = ufm_amd.ufm_base_config(resolution_wh=(res, res))
    variant = []
    if args.info_sharing == "cross_attention":  # same width / depth / heads as the assumed UFM-Base joint-attention trunk
        cfg["info_sharing_str"] = "cross_attention"
        cfg["info_sharing_kwargs"] = dict(name="info_sharing", input_embed_dim=1024, num_views=2, depth=12, dim=768, num_heads=12,
                                          rope_freq=args.rope or None, init_values=None)
        variant.append("cross_attention info sharing" + (f" + RoPE-2D (freq {args.rope:g})" if args.rope else ""))
    if args.head == "moge_conv":  # MoGe's default widths on the four UFM-Base pyramid levels
        cfg["head_type"] = "moge_conv"
        cfg["feature_head_kwargs"] = dict(input_feature_dims=[1024, 768, 768, 768], dim_out=[2], dim_proj=512, dim_upsample=[256, 128, 128], last_conv_channels=32)
        variant.append("moge_conv flow head")
    model = ufm_amd.UniFlowMatchConfidence(**cfg).eval()
    init_weights_(model, seed=0)  # deterministic CPU RNG: identical weights on every rank
    model = model.to(dev).set_numerics(args.numerics)
    model.engine().micro_batches = args.micro_batches
    model.engine().last_layer_view1 = bool(args.last_layer_view1)
    model.engine().group_heads = bool(args.group_heads)
    model.engine().conv_splitk = bool(args.conv_splitk)
    if args.concurrent_heads >= 0:
        model.engine().concurrent_heads = bool(args.concurrent_heads)

    B = args.batch
    # Global batch = world x B pairs, generated identically on every rank (one seeded CPU stream); each rank computes
    # its contiguous shard and ONE RCCL all_gather per step closes it (ufm_amd.dist.ShardedPredictor: the class the
    # gloo world_size 2/3 CPU tests and the nccl GPU test exercise).  The gather is asynchronous over a ring of two
    # buffers: step i's gather (xGMI) overlaps step i+1's compute, and every gather is waited for inside the timed
    # region (result() of the previous step each step, drain() before the closing fence).
    g = torch.Generator().manual_seed(1234)
    src = torch.randint(0, 256, (world * B, res, res, 3), dtype=torch.uint8, generator=g).to(dev)
    tgt = torch.randint(0, 256, (world * B, res, res, 3), dtype=torch.uint8, generator=g).to(dev)
    sharded = None
    if use_dist:
        from ufm_amd.dist import ShardedPredictor, shard_bounds

        def predict_pair_batch(s_, t_):
            o_ = model.predict_correspondences_batched(s_, t_)
            return o_.flow.flow_output, o_.covisibility.mask

        sharded = ShardedPredictor(predict_pair_batch, depth=2)
    last = [None]

    def step():
        if sharded is None:
            return model.predict_correspondences_batched(src, tgt)
        tk = sharded.submit(src, tgt)
        if last[0] is not None:
            # the previous step's gathered results are complete and every rank's status row is clean (check=True: a rank
            # whose predict raised makes ALL ranks raise here, before the next collective)
            sharded.wait(last[0])
        last[0] = tk
        return tk

    def fence():
        if use_dist:
            sharded.drain()
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        step()
        ev[i + 1].record()
    fence()
    elapsed = time.perf_counter() - t0
    per_rank = None
    if use_dist:
        mine_t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        all_t = torch.empty(world, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(all_t, mine_t)
        per_rank = [1e3 * float(v) / args.steps for v in all_t.tolist()]
        elapsed = float(all_t.max().item())  # MAX over ranks
    step_ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps))
    p50 = step_ms[len(step_ms) // 2]

    line = {
        "metric": "image-pairs/sec, UFM-Base 518x518",
        "value": world * B * args.steps / elapsed,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "p50_latency_ms": p50,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": {"fast": "bf16", "precise": "bf16x3", "parity": "f32"}[args.numerics],
        "data": "synthetic",
        "config": {
            "workload": f"UFM-Base (DINOv2 ViT-L/14 + 12x768 joint-attention + 2 DPT heads), random-init weights, "
                        f"batch={B} {res}x{res} synthetic uint8 pairs per GPU, predict_correspondences_batched end to end",
            "pairs_per_gpu": B,
            "global_batch": world * B,
            "resolution": res,
            "numerics": f"{args.numerics}: " + {"fast": "bf16 MFMA trunk (fp32 accumulate/residual/LN/softmax stats), DPT heads in bf16x3 split precision (~2^-17 rel., fp32 accumulate)",
                                                 "precise": "bf16x3 split precision (hi*hi + hi*lo + lo*hi, fp32 accumulate) for every contraction: trunk GEMMs, attention, heads",
                                                 "parity": "fp32 MFMA everywhere"}[args.numerics],
            "parallelism": f"dp{world} (pair-batch split; one async double-buffered RCCL all_gather of the results per step, ufm_amd.dist.ShardedPredictor)",
            "micro_batches_per_gpu": args.micro_batches,
        },
    }
    if share_gpu:
        line["config"]["rehearsal"] = f"{world} ranks SHARING one GPU over gloo: functional rehearsal of the N > 1 path, not a measurement"
        line["metric"] += " [REHEARSAL: ranks share one GPU]"
    if variant:  # NOT the headline configuration: a side measurement of a SURVEY 8(f)4 variant
        line["config"]["variant"] = ", ".join(variant)
        line["metric"] += " [variant: " + ", ".join(variant) + "]"

    # ---- N > 1: the gathered result of ANOTHER rank's shard equals this rank's own recomputation, bit for bit ----
    if use_dist:
        gms = sorted(sharded.gather_ms[-args.steps:]) or [0.0]
        line["per_rank_ms_per_step"] = {"min": min(per_rank), "max": max(per_rank), "all": [round(v, 3) for v in per_rank]}
        # own shard packed -> every rank's results present on this rank (rank 0's view): rank skew + the xGMI transfer
        line["gather_wait_ms"] = {"p50": gms[len(gms) // 2], "max": gms[-1]}
        line["host_threads_per_rank"] = {"launch_threads": args.micro_batches, "torch_intra_op": torch.get_num_threads()}
        flow_all, mask_all = sharded.result(last[0])
        lo, _ = shard_bounds(world * B, (rank + 1) % world, world)
        mine = model.predict_correspondences_batched(src[lo : lo + 1], tgt[lo : lo + 1])
        same = bool(torch.equal(mine.flow.flow_output, flow_all[lo : lo + 1]) and torch.equal(mine.covisibility.mask, mask_all[lo : lo + 1]))
        ok = torch.tensor([1 if same else 0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        line["gather_check"] = {"bitwise_equal_to_local_recompute": bool(ok.item()), "pairs_gathered": int(flow_all.shape[0])}
        if not bool(ok.item()):
            # reported in the JSON line (and loudly here) rather than raised: a failed check must not also lose the timing
            print(f"[bench rank {rank}] WARNING: gathered results differ from the local recomputation (max-abs flow diff "
                  f"{(mine.flow.flow_output - flow_all[lo : lo + 1]).abs().max().item():.3g})", file=sys.stderr, flush=True)

    src, tgt = src[:B], tgt[:B]  # everything below is single-GPU work on one shard-sized batch
    # ---- per-kernel durations: three extra instrumented single-stream steps, HIP events on the launch stream, per-launch median ----
    if rank == 0 and not args.no_kernel_timing:
        summ, records = instrumented_steps(lambda: model.predict_correspondences_batched(src, tgt))
        kernels = {}
        for name, d in summ.items():
            work = sum(meta_work(m) for m in d["metas"])
            entry = {"launches": d["launches"], "ms_per_step": d["ms"], "avg_launch_us": 1e3 * d["ms"] / d["launches"]}
            if name in MFMA_PEAKS:
                peak = MFMA_PEAKS[name]
                entry.update(bound="mfma", algorithmic_gflop=work / 1e9, achieved=work / (d["ms"] * 1e-3) / 1e12, peak=peak, unit="TFLOP/s")
                entry["frac"] = entry["achieved"] / peak
                if name in ("ufm_gemm_bf16", "ufm_gemm_bf16x3", "ufm_conv2d_nhwc_bf16x3"):
                    entry["per_shape"] = per_shape_table([r for r in records if r[0] == name], peak)
            elif work:
                entry.update(bound="hbm", algorithmic_gb=work / 1e9, achieved=work / (d["ms"] * 1e-3) / 1e9, peak=PEAK_HBM_GBS, unit="GB/s")
                entry["frac"] = entry["achieved"] / PEAK_HBM_GBS
            kernels[name] = entry
        mf = {k: v for k, v in kernels.items() if v.get("bound") == "mfma"}
        dom = max(mf, key=lambda k: mf[k]["ms_per_step"])
        d = mf[dom]
        # HBM traffic per launch: PMC counters cannot be read from inside the process; they come from the committed
        # summary of `tools/pmc_traffic.sh` (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate --pmc passes over
        # this same command), averaged over the launches of the family like `achieved`.
        pmc, traffic_src = {}, None
        # the newest committed summary (profiles/rNN/pmc_step_summary.json): a round's counters are regenerated on its final tree
        pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_step_summary.json")))
        pmc_path = pmc_files[-1] if pmc_files else os.path.join(ROOT, "profiles", "r03", "pmc_step_summary.json")
        pmc_rel = os.path.relpath(pmc_path, ROOT)
        if os.path.exists(pmc_path) and B == 8 and res == 518 and args.numerics == "fast":
            cand = json.load(open(pmc_path))
            meta = cand.get("_meta", {})
            # counters measured on other kernel sources are not evidence for this run: drop them (traffic = null)
            if meta.get("csrc_sha256") == csrc_sha256():
                pmc, traffic_src = cand, {"file": pmc_rel, "csrc_sha256": meta["csrc_sha256"], "commit": meta.get("commit_at_summarise_time")}
            else:
                traffic_src = {"file": pmc_rel, "stale": True, "measured_on_csrc_sha256": meta.get("csrc_sha256"), "this_tree": csrc_sha256()}
        for k, v in kernels.items():
            if k in pmc and "hbm_bytes_per_step" in pmc[k]:
                v["traffic"] = pmc[k]["hbm_bytes_per_step"] / v["launches"]  # per C-ABI call, like `achieved`
                v["mfma_busy_frac_pmc"] = pmc[k].get("mfma_busy_frac")
        line["roofline"] = {
            "kernel": dom, "bound": "mfma", "achieved": d["achieved"], "peak": d["peak"], "unit": "TFLOP/s", "frac": d["frac"],
            "traffic": d.get("traffic"), "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic_src,
            "avg_launch_us": d["avg_launch_us"], "algorithmic_gflop_per_launch": d["algorithmic_gflop"] / d["launches"],
            "measured_on": "three single-stream instrumented steps (HIP events around every launch, on the launch stream), per-launch medians; launches on a stream the "
                           "engine has not flagged with ufm_hint_concurrent_stream are dispatched for their own latency, the timed two-stream steps for CU time",
        }
        attn_name = next((k for k in ("ufm_attention_bf16", "ufm_attention_bf16x3", "ufm_attention_f32") if k in kernels), None)
        if attn_name:
            a = kernels[attn_name]
            line["attention"] = {"achieved": a["achieved"], "peak": a["peak"], "unit": "TFLOP/s", "frac": a["frac"], "ms_per_step": a["ms_per_step"]}
        line["kernels"] = kernels
        line["instrumented_step_ms"] = sum(v["ms_per_step"] for v in kernels.values())
        if args.micro_batches > 1 and B >= 4:
            line["pipeline_kernels"] = pipeline_families(model, src, tgt, line["ms_per_step"])
            line["roofline"]["dispatch_policy"] = "latency (single-stream instrumented leg); the timed two-stream steps run the CU-time policy: see pipeline_kernels"

    # ---- CPU baseline: the oracle on this box's host cores, bounded sample ----
    ref_oracle = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ufm_ref as R

        ncores = host_cores()
        torch.set_num_threads(ncores)
        oracle = R.UFMRef(**R.ufm_base_config(resolution_wh=(res, res))).eval()
        oracle.load_state_dict(model.state_dict(), strict=True)
        s1, t1 = src[:1].cpu(), tgt[:1].cpu()
        tiny = R.UFMRef(**R.ufm_tiny_config()).eval()  # page torch's CPU kernels in, outside the timing
        tiny.predict_correspondences_batched(torch.zeros(1, 56, 56, 3, dtype=torch.uint8), torch.zeros(1, 56, 56, 3, dtype=torch.uint8))
        times1 = []
        for _ in range(2):  # B = 1, two timed runs (the tiny model above was the warm-up of torch's CPU kernels)
            c0 = time.perf_counter()
            ref = oracle.predict_correspondences_batched(s1, t1)
            times1.append(time.perf_counter() - c0)
        c0 = time.perf_counter()
        oracle.predict_correspondences_batched(src[:2].cpu(), tgt[:2].cpu())  # B = 2, one timed run
        t_b2 = time.perf_counter() - c0
        cpu_s = min(times1)
        cpu_model = ""
        try:
            for ln in open("/proc/cpuinfo"):
                if ln.startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        got = model.predict_correspondences_batched(src[:1], tgt[:1])
        line["cpu_baseline"] = {
            "value": 1.0 / cpu_s, "unit": "pairs/s", "cores": ncores, "kind": "port", "cpu_model": cpu_model,
            "latency_s_b1_min": min(times1), "latency_s_b1_max": max(times1), "pairs_per_s_b2": 2.0 / t_b2,
            "sample": f"the same workload ({res}x{res}, same weights), fp32 eager-PyTorch oracle on {ncores} threads: 2 timed runs of 1 pair "
                      f"({times1[0]:.1f} s, {times1[1]:.1f} s; value = best) + 1 run of 2 pairs ({t_b2:.1f} s)",
        }
        ref_oracle = ref
        line["check_vs_oracle"] = {
            "numerics": args.numerics,
            "flow_max_abs": float((got.flow.flow_output.cpu() - ref.flow.flow_output).abs().max()),
            "flow_range": float(ref.flow.flow_output.abs().max()),
            "covis_max_abs": float((got.covisibility.mask.cpu() - ref.covisibility.mask).abs().max()),
        }

    # ---- single-pair latency (BASELINE metric: "pairs/s + p50 latency"): wall clock of one synchronous call ----
    if rank == 0 and world == 1 and not args.no_latency:  # (a one-GPU measurement; at N > 1 the graph capture would run beside live RCCL work)
        s1d, t1d = src[:1].contiguous(), tgt[:1].contiguous()
        eager = p50_ms(lambda: model.predict_correspondences_batched(s1d, t1d), 20, 3)
        lat = {"eager_p50": eager, "iters": 20, "batch": 1}
        try:
            gp = ufm_amd.GraphedPredictor(model, s1d, t1d)
            lat["graph_replay_p50"] = p50_ms(lambda: gp(s1d, t1d), 20, 3)
            a_ = model.predict_correspondences_batched(s1d, t1d).flow.flow_output.clone()
            lat["graph_bitwise_equals_eager"] = bool(torch.equal(gp(s1d, t1d).flow.flow_output, a_))
        except Exception as exc:  # reported, not fatal: the eager number stands on its own
            lat["graph_error"] = repr(exc)[:200]
        line["latency_b1_ms"] = lat

    # ---- the same workload in numerics "precise" (bf16x3 split precision everywhere: the 1e-3 px gate on the bf16 matrix cores) ----
    if rank == 0 and world == 1 and args.numerics == "fast" and not args.no_precise_mode:
        model.set_numerics("precise")
        model.engine().micro_batches = args.micro_batches
        for _ in range(2):
            model.predict_correspondences_batched(src, tgt)
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        for _ in range(5):
            model.predict_correspondences_batched(src, tgt)
        torch.cuda.synchronize()
        pre_s = (time.perf_counter() - c0) / 5
        pm = {"value": B / pre_s, "unit": "pairs/s", "ms_per_step": 1e3 * pre_s, "steps": 5, "dtype": "bf16x3",
              "numerics": "precise: hi*hi + hi*lo + lo*hi on the bf16 matrix cores for every contraction (trunk GEMMs, attention, heads), fp32 accumulate / residual / LN / softmax statistics"}
        if ref_oracle is not None:
            gotp = model.predict_correspondences_batched(src[:1], tgt[:1])
            pm["flow_max_abs"] = float((gotp.flow.flow_output.cpu() - ref_oracle.flow.flow_output).abs().max())
            pm["covis_max_abs"] = float((gotp.covisibility.mask.cpu() - ref_oracle.covisibility.mask).abs().max())
        if not args.no_kernel_timing:
            summ, records = instrumented_steps(lambda: model.predict_correspondences_batched(src, tgt))
            pk = {}
            for name, d in summ.items():
                if name in MFMA_PEAKS:
                    work = sum(meta_work(m) for m in d["metas"])
                    pk[name] = {"launches": d["launches"], "ms_per_step": d["ms"], "achieved": work / (d["ms"] * 1e-3) / 1e12, "peak": MFMA_PEAKS[name],
                                "unit": "TFLOP/s (algorithmic)", "frac": work / (d["ms"] * 1e-3) / 1e12 / MFMA_PEAKS[name]}
                    if name == "ufm_gemm_bf16x3":
                        pk[name]["per_shape"] = per_shape_table([r for r in records if r[0] == name], MFMA_PEAKS[name])
                else:
                    pk[name] = {"launches": d["launches"], "ms_per_step": d["ms"]}
            pm["kernels"] = pk
        line["precise_mode"] = pm
        model.set_numerics("fast")

    # ---- the same workload in numerics "parity" (exact-fp32 MFMA everywhere: the mode that meets the 1e-3 px gate) ----
    if rank == 0 and world == 1 and args.numerics == "fast" and not args.no_parity_mode:
        model.set_numerics("parity")
        model.engine().micro_batches = args.micro_batches
        model.predict_correspondences_batched(src, tgt)
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        for _ in range(3):
            model.predict_correspondences_batched(src, tgt)
        torch.cuda.synchronize()
        par_s = (time.perf_counter() - c0) / 3
        pm = {"value": B / par_s, "unit": "pairs/s", "ms_per_step": 1e3 * par_s, "steps": 3, "dtype": "f32",
              "numerics": "parity: exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) for every contraction"}
        if ref_oracle is not None:
            gotp = model.predict_correspondences_batched(src[:1], tgt[:1])
            pm["flow_max_abs"] = float((gotp.flow.flow_output.cpu() - ref_oracle.flow.flow_output).abs().max())
            pm["covis_max_abs"] = float((gotp.covisibility.mask.cpu() - ref_oracle.covisibility.mask).abs().max())
        line["parity_mode"] = pm
        model.set_numerics("fast")

    # ---- BASELINE configs 4 and 5 under the same clock (side legs, rank 0, N = 1; never part of `value`) ----
    if rank == 0 and world == 1 and args.numerics == "fast" and not args.no_side_configs and B == 8 and res == 518:
        del model
        torch.cuda.empty_cache()
        line["config4"] = side_config(ufm_amd, hip, "refine", 518, 8, steps=6, micro_batches=args.micro_batches)
        line["config5"] = side_config(ufm_amd, hip, "base", 1036, 2, steps=4, micro_batches=args.micro_batches)
        line["default_res"] = side_config(ufm_amd, hip, "default_res", 518, 8, steps=6, micro_batches=args.micro_batches)

    # ---- the clock the chip holds under the dominant kernel family (in-kernel stamps, diagnostic instantiations) ----
    if rank == 0 and world == 1 and args.numerics == "fast" and not args.no_clock and "roofline" in line:
        clk = gemm_clock_under_load(hip)
        line["roofline"]["clock_ghz"] = clk["clock_ghz"]
        line["roofline"]["frac_at_clock"] = line["roofline"]["achieved"] / (line["roofline"]["peak"] * clk["clock_ghz"] / 2.4)
        line["roofline"]["clock_source"] = clk["how"]
        cv = line.get("kernels", {}).get("ufm_conv2d_nhwc_bf16x3")
        if cv is not None and clk.get("conv_clock_ghz"):
            cv["clock_ghz"] = clk["conv_clock_ghz"]
            cv["frac_at_clock"] = cv["achieved"] / (cv["peak"] * clk["conv_clock_ghz"] / 2.4)

    if rank == 0:
        print(json.dumps(order_line(line, B)))
    if use_dist:
        dist.barrier()  # rank 0's single-rank extras (kernel timing, latency) are done: all ranks leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
