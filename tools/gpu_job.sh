#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/job_vae_prof -- python3 $R/tools/vae_bench.py --n 16 --iters 3 > $O/job_vae_prof.log 2>&1 || exit 1
find $O/job_vae_prof -name "*kernel_trace.csv" -size +20M -delete
tail -1 $O/job_vae_prof.log
