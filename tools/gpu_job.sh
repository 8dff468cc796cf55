#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_vae.py -m gpu -x -q --timeout 500 2>&1 | tail -3 || exit 1
{ python tools/sd_bench.py --scorer brightness; python tools/sd_bench.py --host-preprocess; python tools/sd_bench.py;
  DTS_DIST_BACKEND=gloo python tools/sd_bench.py --gpus 2; } 2>&1 | grep -v amdgpu | grep "^SD beam\|Error\|error" > $O/job_sd.txt
cat $O/job_sd.txt
