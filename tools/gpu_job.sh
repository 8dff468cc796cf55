#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
B="--steps 40 --warmup 3 --no-cpu-baseline --no-e2e --no-parity --no-kernel-timing --no-subrecords"
run() { echo "$1: $(env $2 python bench.py $B $3 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"; }
BASE=DTS_LIB_PATH=$R/diffusion_tts_amd/libdts_hip_base.so
for i in 1 2 3; do
run "c64 new" "X=1" "--candidates 64"
run "c64 base" "$BASE" "--candidates 64"
done
env $BASE python tools/conv_bench.py --n 64 --stats --variants conv_variant=1 > $O/job_cb64_base.txt 2>&1
python tools/conv_bench.py --n 64 --stats --variants conv_variant=1 > $O/job_cb64_new.txt 2>&1
tail -1 $O/job_cb64_base.txt; tail -1 $O/job_cb64_new.txt
