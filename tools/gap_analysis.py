#!/usr/bin/env python3
"""Inter-kernel gaps from a rocprofv3 --kernel-trace CSV: how much of the step is the GPU idle between kernels?"""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50], int(r.get("Stream_Id", 0) or 0)) for r in csv.DictReader(open(f))]
rows.sort()
n = len(rows)
# take the last third of the trace (steady-state steps)
rows = rows[2 * n // 3:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_end, gaps = 0, rows[0][0], []
for s, e, name, _ in rows:
    if s > cur_end:
        gaps.append((s - cur_end, name))
        busy += e - s
        cur_end = e
    else:
        busy += max(0, e - cur_end)
        cur_end = max(cur_end, e)
wall = t1 - t0
print(f"kernels {len(rows)}  wall {wall/1e6:.2f} ms  busy(union) {busy/1e6:.2f} ms  idle {100*(wall-busy)/wall:.1f}%  mean gap {sum(g for g,_ in gaps)/max(1,len(gaps))/1e3:.2f} us over {len(gaps)} gaps")
by = collections.defaultdict(lambda: [0, 0])
for g, name in gaps:
    by[name][0] += 1; by[name][1] += g
for name, (c, tot) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  gap before {name:50s} x{c:4d}  total {tot/1e3:8.1f} us  mean {tot/c/1e3:.2f} us")
