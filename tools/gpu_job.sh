#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
B="--steps 20 --warmup 2 --no-cpu-baseline --no-e2e --no-parity --no-kernel-timing --no-subrecords"
run() { echo "$1: $(env $2 python bench.py $B $3 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"; }
run "c64 unbranched" "X=1" "--candidates 64"
run "c64 branch rows 32 (2 branches)" "DTS_BRANCH_ROWS=32" "--candidates 64"
run "c64 branch rows 16 (4 branches)" "DTS_BRANCH_ROWS=16" "--candidates 64"
run "c64 unbranched" "X=1" "--candidates 64"
