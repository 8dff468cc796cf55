#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_search.py tests/test_gpu_fullsize.py -x -q -k "attention or ddpmpp or config1 or config2 or denoiser" > $O/job_att.log 2>&1 || { tail -40 $O/job_att.log; exit 1; }
tail -1 $O/job_att.log
B="--workload ddpmpp32_rejection --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-timing"
run() { echo "$1: $(env $2 python bench.py $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"; }
for i in 1 2 3; do
run "ddpm32 regs (DTS_ATT_DB=2)" "DTS_ATT_DB=2"
run "ddpm32 dma" "X=1"
done
