#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "group_norm_of_its_input or ping_pong or repeat_bit" 2>&1 | tail -3 || exit 1
B="--steps 20 --warmup 2 --no-cpu-baseline --no-e2e --no-parity --no-kernel-timing --no-subrecords"
run() { echo "$1: $(env $2 python bench.py $B $3 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"; }
run "c64 plain" "X=1" "--candidates 64"
run "c64 gn fused" "DTS_GN_FUSE=1" "--candidates 64"
run "c64 plain" "X=1" "--candidates 64"
run "c64 gn fused" "DTS_GN_FUSE=1" "--candidates 64"
run "c8 plain" "X=1" "--candidates 8"
run "c8 gn fused" "DTS_GN_FUSE=1" "--candidates 8"
run "c16 plain" "X=1" "--candidates 16"
run "c16 gn fused" "DTS_GN_FUSE=1" "--candidates 16"
