#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
B="--steps 20 --warmup 2 --no-cpu-baseline --no-e2e --no-parity --no-kernel-timing --no-subrecords"
run() { echo "$1: $(env $2 python bench.py $B $3 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"; }
run "c64 old lib" "DTS_LIB_PATH=$R/diffusion_tts_amd/libdts_hip_old.so" "--candidates 64"
run "c64 new" "X=1" "--candidates 64"
run "c8 old lib" "DTS_LIB_PATH=$R/diffusion_tts_amd/libdts_hip_old.so" "--candidates 8"
run "c8 new" "X=1" "--candidates 8"
run "c16 old lib" "DTS_LIB_PATH=$R/diffusion_tts_amd/libdts_hip_old.so" "--candidates 16"
run "c16 new" "X=1" "--candidates 16"
timeout -k 10 1000 python -m pytest tests/ -m gpu -x -q --timeout 900 > $O/job_pytest.log 2>&1; rc=$?
tail -4 $O/job_pytest.log; exit $rc
