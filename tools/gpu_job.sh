#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
for n in 64 32; do
timeout -k 10 500 python tools/conv_tune.py --n $n --iters 5 > $O/job_tune_n$n.txt 2>&1 || { tail -20 $O/job_tune_n$n.txt; exit 1; }
tail -1 $O/job_tune_n$n.txt
done
