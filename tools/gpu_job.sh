#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python tools/mcts_profile.py --S 64 2>&1 | grep -v amdgpu.ids | tee $O/job_mcts_profile.txt
