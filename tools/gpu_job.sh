#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
Q="--no-cpu-baseline --no-e2e --no-parity --no-subrecords --no-kernel-timing"
for v in 1 0 1 0; do
DTS_X3_FUSE_IMAGES=$v timeout -k 10 300 python bench.py --steps 10 --warmup 2 --dtype f16x3 $Q 2> $O/job_bench_x3.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f16x3 fuse_images=$v', 'ms/step', d['ms_per_step'], 'evals/s', d['value'])" || { tail -20 $O/job_bench_x3.err; exit 1; }
done
