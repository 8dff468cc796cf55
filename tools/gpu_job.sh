#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_search.py -m gpu -x -q -s -k "throughput_modes" --timeout 900 > $O/job_pytest.log 2>&1; rc=$?
grep -v amdgpu $O/job_pytest.log | tail -25; exit $rc
