#!/bin/bash
# same-box sweep of the engine's concurrency switches (bench.py value only); output: one line per configuration
R=${GRAFT_REPO_ROOT:-/root/repo}
F="--steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --no-precise-mode --no-latency --no-kernel-timing"
for cfg in "--micro-batches 2" "--micro-batches 2 --concurrent-heads 1" "--micro-batches 1" "--micro-batches 1 --concurrent-heads 0" "--micro-batches 3" "--micro-batches 2"; do
  v=$(python3 $R/bench.py $F $cfg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), round(d['ms_per_step'],3))")
  echo "$cfg : $v"
done
