#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
B="--steps 30 --warmup 3 --no-cpu-baseline --no-e2e --no-parity --no-kernel-timing --no-subrecords"
run() { echo "$1: $(env $2 python bench.py $B $3 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"; }
for c in 16 32 64; do
run "c$c default" "X=1" "--candidates $c"
run "c$c half-round(mt4)" "DTS_EXP_HALF_ROUND=1" "--candidates $c"
run "c$c default" "X=1" "--candidates $c"
run "c$c half-round(mt4)" "DTS_EXP_HALF_ROUND=1" "--candidates $c"
done
B="--workload ddpmpp32_rejection --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-timing"
for i in 1 2; do
run "ddpm32 default" "X=1" ""
run "ddpm32 half-round(mt4)" "DTS_EXP_HALF_ROUND=1" ""
done
