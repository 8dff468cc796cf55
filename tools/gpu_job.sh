#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
for v in 100 101 102 104 103 106 107; do
  python tools/conv_stamps.py --n 8 --variant $v 2>&1 | grep "1x1\|L3 3x3\|L2 3x3 576" | cut -c1-215 > $O/job_diag_$v.txt
  echo "== variant $v"; cat $O/job_diag_$v.txt
done
