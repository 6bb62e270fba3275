#!/bin/bash
# LDS bank-conflict cycles of ufm_dpt_tail_fused with the half-swapped (default) and the plain stage-A tile (tools/lab/tail_ab.py)
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-pmc_tail}
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O -- python3 $R/tools/lab/tail_ab.py > /dev/null 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
order = []
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "dpt_tail_fused" in r["Kernel_Name"]:
            rows[r["Dispatch_Id"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
# tail_ab.py alternates: 6 launches of variant 1 (1 + 5), then 6 of variant 3, ...
ids = sorted(rows, key=int)
agg = {1: collections.defaultdict(list), 3: collections.defaultdict(list)}
for i, d in enumerate(ids):
    v = 1 if (i // 6) % 2 == 0 else 3
    for c, vals in rows[d].items():
        agg[v][c].append(sum(vals))
for v, name in ((1, "half-swapped T (round 5)"), (3, "plain T (rounds 1-4)")):
    a = {c: sum(x) / len(x) for c, x in agg[v].items()}
    print(name, {c: round(x / 1e6, 2) for c, x in a.items()}, "conflict share of LDS cycles %.3f" % (a["SQ_LDS_BANK_CONFLICT"] / a["SQ_LDS_IDX_ACTIVE"]))
PY
