#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_search.py -m gpu -q --timeout 300 -s -k "split_precision or conv2d or search_parity or denoiser_forward or classifier_and" > $O/job_x3.log 2>&1 || { grep -v "^split precision" $O/job_x3.log | tail -40; grep "^split precision" $O/job_x3.log; exit 1; }
grep "^split precision" $O/job_x3.log; tail -2 $O/job_x3.log
