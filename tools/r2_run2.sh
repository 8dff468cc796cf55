#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2b; mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?"; cat $O/bench_default.json; tail -3 $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof8 -- python3 $R/bench.py --steps 5 --warmup 1 --candidates 8 --no-cpu-baseline --no-kernel-timing --no-e2e > $O/prof8.log 2>&1 || tail -5 $O/prof8.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof64 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-e2e > $O/prof64.log 2>&1 || tail -5 $O/prof64.log
find $O -name "*kernel_stats.csv" | head; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; du -sh $O
