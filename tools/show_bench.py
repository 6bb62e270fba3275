#!/usr/bin/env python3
"""Pretty-print a bench.py JSON line (per-kernel breakdown)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("pairs/s", round(d["value"], 2), "ms/step", round(d["ms_per_step"], 3), "p50", round(d["p50_latency_ms"], 3))
for k, v in sorted(d.get("kernels", {}).items(), key=lambda kv: -kv[1]["ms_per_step"]):
    print(f"{k:32s} {v['launches']:4d} {v['ms_per_step']:8.3f} ms  frac={v.get('frac', 0):.3f} ach={v.get('achieved', 0):.1f} {v.get('unit', '')}")
print("instrumented_step_ms", d.get("instrumented_step_ms"))
for k in ("roofline", "attention", "check_vs_oracle", "cpu_baseline", "parity_mode", "latency_b1_ms"):
    if k in d:
        print(k, d[k])
