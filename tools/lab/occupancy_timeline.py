#!/usr/bin/env python3
"""Estimated CU fill of the real (multi-stream) pipeline from a rocprofv3 --kernel-trace CSV.
For every kernel: workgroups = grid / block; residency per CU from its LDS / VGPR / block size (what the hardware allows); its DEMAND =
min(1, workgroups / (256 CUs x residency)) of the chip while it runs.  The timeline sums the demands of the kernels running at each instant
(capped at 1); 1 - that is CU time nobody asked for.  Reported per step of the last steps of the trace: wall, integral of the fill, and
the under-filled time attributed to the kernels that were running (share of (1 - fill) x dt split by demand).
   python tools/lab/occupancy_timeline.py <dir with *_kernel_trace.csv> [steps_to_skip_fraction]"""
import csv, glob, sys, collections

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
NCU = 256
rows = []
for r in csv.DictReader(open(f)):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    nwg = max(1, grid // max(1, wg))
    lds = int(r["LDS_Block_Size"])
    regs = int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"])
    waves = (wg + 63) // 64
    per_simd_waves = max(1, 512 // max(regs, 1))          # waves a SIMD can hold at this register count
    res_regs = max(1, (per_simd_waves * 4) // waves)      # workgroups per CU by registers
    res_lds = max(1, (160 * 1024) // lds) if lds else 32
    res_waves = max(1, 32 // waves)                       # 8 waves per SIMD x 4
    res = max(1, min(res_regs, res_lds, res_waves))
    name = r["Kernel_Name"]
    short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:62]
    rows.append((s, e, short, nwg, res, min(1.0, nwg / (NCU * res))))
rows.sort()
# window: whole steps of the bench's timed loop -- a step starts with its patchify launches (a burst of them within 1 ms); skip the first
# `skip` steps (warm-up), end at the start of the last step
starts = []
for s, e, short, *_ in rows:
    if "patchify" in short and (not starts or s - starts[-1] > 5e6):
        starts.append(s)
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t_lo, t_hi = starts[skip], starts[-1]
nsteps = len(starts) - 1 - skip
rows = [r for r in rows if r[0] >= t_lo and r[0] < t_hi]
print(f"{nsteps} steps, {(t_hi - t_lo) / nsteps / 1e6:.2f} ms per step")
ev = []
for i, (s, e, *_r) in enumerate(rows):
    ev.append((s, 1, i))
    ev.append((e, 0, i))
ev.sort()
active = set()
under = collections.defaultdict(float)
alone = collections.defaultdict(float)
fill_int = 0.0
idle = 0.0
t_prev = ev[0][0]
for t, kind, i in ev:
    dt = t - t_prev
    if dt > 0:
        dem = sum(rows[j][5] for j in active)
        fill = min(1.0, dem)
        fill_int += fill * dt
        if not active:
            idle += dt
        elif fill < 1.0:
            for j in active:
                under[rows[j][2]] += (1.0 - fill) * dt * rows[j][5] / dem
            if len(active) == 1:
                alone[rows[next(iter(active))][2]] += dt
    t_prev = t
    if kind:
        active.add(i)
    else:
        active.discard(i)
wall = ev[-1][0] - ev[0][0]
print(f"{len(rows)} kernels, wall {wall / 1e6:.2f} ms, estimated CU fill {fill_int / wall:.3f}, no kernel at all {idle / wall:.3f} ({idle / 1e6 / nsteps:.3f} ms/step)")
print("under-filled CU time by running kernel (ms of whole-chip time per step; share of the wall):")
for k, v in sorted(under.items(), key=lambda kv: -kv[1])[:22]:
    print(f"  {k:62s} {v / 1e6 / nsteps:7.3f} ms/step  {v / wall:6.3f}   ran alone {alone.get(k, 0) / 1e6 / nsteps:7.3f} ms/step")
