#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python tools/att512_bench.py --n 16 2>&1 | grep -v amdgpu
python tools/att512_bench.py --n 2 2>&1 | grep -v amdgpu
