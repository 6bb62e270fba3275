#!/bin/bash
# SQ counters of the two split-precision attention kernels (tools/lab/attn_x3_ab.py), separate --pmc passes, --kernel-trace only.
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-pmc_attn_x3}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -- python3 $R/tools/lab/attn_x3_ab.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/b -- python3 $R/tools/lab/attn_x3_ab.py > /dev/null 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "attn_x3" in r["Kernel_Name"]:
            key = ("round5 " if "pw" in r["Kernel_Name"] else "round1 ") + r["Grid_Size"]
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    a = {c: sum(v) / len(v) for c, v in acc[k].items()}
    wc = a.get("SQ_WAVE_CYCLES", 1)
    print(k, {c: round(v / 1e6, 2) for c, v in a.items()})
    if "SQ_WAIT_ANY" in a:
        print("   of wave cycles: wait_any %.3f  wait_inst_any %.3f  active_inst_any %.3f | mfma busy of GUI active %.3f" % (
            a["SQ_WAIT_ANY"] / wc, a["SQ_WAIT_INST_ANY"] / wc, a["SQ_ACTIVE_INST_ANY"] / wc, a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * a["GRBM_GUI_ACTIVE"] / 8)))
PY
