#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/ -m gpu -x -q --timeout 900 > $O/job_pytest.log 2>&1; rc=$?
tail -5 $O/job_pytest.log; exit $rc
