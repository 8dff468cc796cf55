#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/vae_prof -- python3 $R/tools/vae_bench.py --n 16 > $O/vae_prof.log 2>&1; echo "rc=$?"
find $O -name "*kernel_trace.csv" -size +20M -delete
f=$(find $O/vae_prof -name "*kernel_stats.csv" | head -1); head -25 $f | cut -d, -f1-5 | cut -c1-200
