#!/usr/bin/env python3
"""Summarise tools/pmc_traffic.sh output into profiles/<round>/pmc_step_summary.json: per kernel family, average
per-launch HBM bytes (FETCH_SIZE is in KiB and is doubled: on gfx950 it reports exactly half the bytes of wide
coalesced reads -- MI355X_MICROARCH.md section HBM; WRITE_SIZE is exact for 16-B stores), MFMA-busy fraction, LDS
bank-conflict cycles."""
import collections, csv, glob, hashlib, json, os, subprocess, sys
out_path = sys.argv[1] if len(sys.argv) > 1 else "profiles/r03/pmc_step_summary.json"


def csrc_sha256(root):
    """Content hash of the kernel sources the counters were measured on (bench.py recomputes it: a summary whose hash
    differs from the tree it runs in is stale and its `traffic` is dropped)."""
    h = hashlib.sha256()
    d = os.path.join(root, "ufm_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 3  # tools/pmc_traffic.sh traces --warmup 1 --steps 2: three identical steps
FAM = {"gemm_bf16": "ufm_gemm_bf16", "attn_bf16_kernel": "ufm_attention_bf16", "attn_pw_kernel": "ufm_attention_bf16", "conv_x3": "ufm_conv2d_nhwc_bf16x3",
       "layernorm_kernel": "ufm_layernorm", "upsample": "ufm_upsample_bilinear_nhwc", "dpt_tail_fused": "ufm_dpt_tail_fused", "head_tail_kernel": "ufm_head_tail",
       "patchify": "ufm_patchify", "unmap": "ufm_unmap", "conv_f32": "ufm_conv2d_nhwc_f32"}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
by_grid = collections.defaultdict(lambda: collections.defaultdict(list))  # round 6: (family, kernel template, grid) -> counter -> values: a grid size identifies a GEMM / conv shape
for d in ("pmc_step_fetch", "pmc_step_write", "pmc_step_util"):
    files = sorted(glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv"), key=os.path.getmtime)
    for f in files[-1:]:  # gpurun_out/ keeps earlier calls' files: only the newest pass counts
        for r in csv.DictReader(open(f)):
            fam = next((v for k, v in FAM.items() if k in r["Kernel_Name"]), None)
            if fam:
                acc[fam][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if fam in ("ufm_gemm_bf16", "ufm_conv2d_nhwc_bf16x3"):
                    short = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
                    by_grid[(fam, short, r.get("Grid_Size", r.get("Grid_Size_X", "?")))][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for fam, d in acc.items():
    a = {k: sum(v) / len(v) for k, v in d.items()}
    e = {"launches_sampled": len(next(iter(d.values())))}
    if "FETCH_SIZE" in a:
        e["hbm_read_bytes_per_launch"] = a["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in a:
        e["hbm_write_bytes_per_launch"] = a["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
        e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
        # per step: one C-ABI call can be two kernel launches (hybrid 8-phase + 128-row split) -- bench.py divides this
        # by ITS launch count (C-ABI calls) to get bytes per launch on the same basis as `achieved`
        e["hbm_bytes_per_step"] = (sum(d["FETCH_SIZE"]) * 1024 * 2 + sum(d["WRITE_SIZE"]) * 1024) / STEPS
        e["kernel_launches_per_step"] = len(d["FETCH_SIZE"]) / STEPS
    if "SQ_VALU_MFMA_BUSY_CYCLES" in a and a.get("GRBM_GUI_ACTIVE"):
        e["mfma_busy_frac"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * a["GRBM_GUI_ACTIVE"] / 8)
        e["wave_wait_frac"] = a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"]
        e["lds_bank_conflict_cycles"] = a["SQ_LDS_BANK_CONFLICT"]
    res[fam] = e
# per (kernel, grid) table of the two big families: which launches carry the bytes above the algorithmic ones
tab = []
for (fam, short, grid), d in by_grid.items():
    if "FETCH_SIZE" not in d:
        continue
    n = len(d["FETCH_SIZE"])
    tab.append({"family": fam, "kernel": short, "grid": grid, "launches_per_step": round(n / STEPS, 2),
                "read_mb_per_launch": round(sum(d["FETCH_SIZE"]) / n * 1024 * 2 / 1e6, 2),
                "write_mb_per_launch": round(sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"]) * 1024 / 1e6, 2) if d.get("WRITE_SIZE") else None,
                "read_gb_per_step": round(sum(d["FETCH_SIZE"]) * 1024 * 2 / STEPS / 1e9, 3)})
tab.sort(key=lambda e: -e["read_gb_per_step"])
res["_by_kernel_and_grid"] = tab
try:
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    commit = ""
res["_meta"] = {"csrc_sha256": csrc_sha256(ROOT), "commit_at_summarise_time": commit,
                "command": "tools/pmc_traffic.sh (bench.py --steps 2 --warmup 1 --micro-batches 1, separate --pmc passes)"}
json.dump(res, open(out_path, "w"), indent=1)
for k, v in res.items():
    if k.startswith("_"):
        continue
    print(k, {a: (round(b, 3) if isinstance(b, float) and b < 10 else int(b)) for a, b in v.items()})
print("by kernel and grid (read GB per step, top 16):")
for e in tab[:16]:
    print("  ", e)
