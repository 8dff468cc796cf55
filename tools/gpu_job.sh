#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
B="--steps 30 --warmup 3 --no-cpu-baseline --no-e2e --no-parity --no-kernel-timing --no-subrecords"
run() { echo "$1: $(env $2 python bench.py $B $3 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"; }
for c in 4 8 16; do
run "c$c strips" "DTS_GN_SMALL_HW=0" "--candidates $c"
run "c$c fused hw<=64" "DTS_GN_SMALL_HW=64" "--candidates $c"
run "c$c fused hw<=256" "DTS_GN_SMALL_HW=256" "--candidates $c"
run "c$c fused hw<=1024" "DTS_GN_SMALL_HW=1024" "--candidates $c"
run "c$c strips" "DTS_GN_SMALL_HW=0" "--candidates $c"
done
