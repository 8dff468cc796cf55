#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_search.py -m gpu -q --timeout 300 -s -k "attention_split or search_parity or denoiser_forward or classifier_and" > $O/job_x3.log 2>&1 || { grep -a "split-precision attention n=" $O/job_x3.log | grep -v print | cut -c1-200; tail -30 $O/job_x3.log | cut -c1-300; exit 1; }
grep -a "split-precision attention n=" $O/job_x3.log | grep -v print | sed 's/^[.F]*//' | cut -c1-200; tail -1 $O/job_x3.log
Q="--no-cpu-baseline --no-e2e --no-parity --no-subrecords --no-kernel-timing"
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --dtype f16x3 $Q 2> $O/job_bench_x3.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f16x3', 'ms/step', d['ms_per_step'], 'evals/s', d['value'])" || { tail -20 $O/job_bench_x3.err; exit 1; }
python tools/att_bench.py --n 64 --x3 2>&1 | grep -v amdgpu | tail -12
