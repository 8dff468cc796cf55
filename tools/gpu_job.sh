#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv or statistics" 2>&1 | tail -3 || exit 1
python tools/conv_bench.py --n 8 --stats --variants conv_variant=0,conv_waves=4 conv_variant=0,conv_waves=8 2>&1 | grep -v amdgpu | tail -15 | cut -c1-200
B="--steps 20 --warmup 2 --no-cpu-baseline --no-e2e --no-parity --no-kernel-timing --no-subrecords"
run() { echo "$1: $(env $2 python bench.py $B $3 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"; }
run "c8 waves=4" "DTS_CONV_WAVES=4" "--candidates 8"
run "c8 auto" "X=1" "--candidates 8"
run "c16 waves=4" "DTS_CONV_WAVES=4" "--candidates 16"
run "c16 auto" "X=1" "--candidates 16"
run "c64 waves=4" "DTS_CONV_WAVES=4" "--candidates 64"
run "c64 auto" "X=1" "--candidates 64"
