#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/final_prof_both
rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_prof_both -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-parity --no-subrecords > $O/final_prof_both.json 2> $O/final_prof_both.err || { tail -5 $O/final_prof_both.err; exit 1; }
find $O/final_prof_both -name "*kernel_trace.csv" -delete; find $O/final_prof_both -name "*.db" -delete
grep "^{" $O/final_prof_both.json | cut -c1-300
