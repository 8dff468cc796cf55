#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
Q="--no-cpu-baseline --no-e2e --no-parity --no-subrecords --no-kernel-timing"
for v in new old new old; do
  if [ $v = old ]; then export DTS_CONV_HALF_ROUND=0; else unset DTS_CONV_HALF_ROUND; fi
  timeout -k 10 300 python bench.py --workload ddpmpp32_rejection --steps 60 --warmup 5 $Q 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ddpmpp32 $v', 'ms/step', d['ms_per_step'], 'evals/s', d['value'])" || exit 1
  timeout -k 10 300 python bench.py --steps 20 --warmup 4 $Q 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('adm64 $v', 'ms/step', d['ms_per_step'], 'evals/s', d['value'])" || exit 1
done
