#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q > $O/job_ops.log 2>&1 || { tail -30 $O/job_ops.log; exit 1; }
tail -1 $O/job_ops.log
B="--steps 30 --warmup 3 --no-cpu-baseline --no-e2e --no-parity --no-kernel-timing --no-subrecords"
run() { echo "$1: $(env $2 python bench.py $B $3 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"; }
BASE=DTS_LIB_PATH=$R/diffusion_tts_amd/libdts_hip_base.so
for c in 4 8 16 32 64; do
run "c$c base" "$BASE" "--candidates $c"
run "c$c new" "X=1" "--candidates $c"
run "c$c base" "$BASE" "--candidates $c"
run "c$c new" "X=1" "--candidates $c"
done
