#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_vae.py tests/test_gpu_sd_shapes.py -x -q > $O/job_vae_tests.log 2>&1 || { tail -30 $O/job_vae_tests.log; exit 1; }
tail -1 $O/job_vae_tests.log
python tools/vae_bench.py --n 16 2>&1 | tail -1 | tee $O/job_vae.txt
python tools/sd_bench.py 2>&1 | tail -1 | tee $O/job_sd.txt
python tools/att512_bench.py --n 16 2>&1 | grep -v amdgpu | tee $O/job_att512.txt
