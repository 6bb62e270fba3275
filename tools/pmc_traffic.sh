#!/bin/bash
# HBM traffic of the kernels of one bench step, per the MI355X guide's recipe: FETCH_SIZE and WRITE_SIZE in
# SEPARATE --pmc passes with --kernel-trace only (never combined with sys/hip traces).  Run on the GPU box:
#   bash tools/pmc_traffic.sh   ->  gpurun_out/pmc_step_{fetch,write}/
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_step_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-parity-mode --no-precise-mode --no-latency --no-side-configs --no-clock --micro-batches 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_step_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-parity-mode --no-precise-mode --no-latency --no-side-configs --no-clock --micro-batches 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_step_util -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-parity-mode --no-precise-mode --no-latency --no-side-configs --no-clock --micro-batches 1 > /dev/null 2>&1
echo done
