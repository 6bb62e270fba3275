#!/bin/bash
# End-of-round evidence on one GPU box (run through gpurun): PMC passes -> summary on this tree -> bench line with live
# traffic -> rocprofv3 kernel stats of the single-stream command.  Outputs under gpurun_out/final/ (copy into profiles/rNN/).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${1:-r05}
O=$R/gpurun_out/final
mkdir -p $O $R/profiles/$ROUND
if [ -z "$SKIP_PMC" ]; then  # SKIP_PMC=1: the summary of this tree is already in profiles/$ROUND (its csrc hash is checked by bench.py)
rm -rf $R/gpurun_out/pmc_step_fetch $R/gpurun_out/pmc_step_write $R/gpurun_out/pmc_step_util
bash $R/tools/pmc_traffic.sh > $O/pmc.log 2>&1
cd $R && python3 tools/pmc_summarize.py profiles/$ROUND/pmc_step_summary.json > $O/pmc_summary.txt 2>&1
cp profiles/$ROUND/pmc_step_summary.json $O/pmc_step_summary.json
fi
python3 bench.py > $O/bench_b8_final.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mb1 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --no-precise-mode --no-latency --no-side-configs --no-clock --micro-batches 1 > $O/bench_b8_final_mb1.json 2> $O/bench_mb1.err
cp $O/prof_mb1/*/*_kernel_stats.csv $O/bench_b8_final_mb1_kernel_stats.csv
rm -rf $O/prof_mb1/*/*_kernel_trace.csv
# the same single-stream command with the two DPT heads serial (--concurrent-heads 0): conv durations not inflated by the
# other head's kernels, so the conv family's `frac` is reproducible from the kept CSV
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mb1_serial -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --no-precise-mode --no-latency --no-side-configs --no-clock --micro-batches 1 --concurrent-heads 0 > $O/bench_b8_final_mb1_serial_heads.json 2> $O/bench_mb1_serial.err
cp $O/prof_mb1_serial/*/*_kernel_stats.csv $O/bench_b8_final_mb1_serial_heads_kernel_stats.csv
rm -rf $O/prof_mb1_serial/*/*_kernel_trace.csv
cd $R && python3 tools/show_bench.py $O/bench_b8_final.json > $O/show.txt 2>&1 || true
head -12 $O/show.txt
