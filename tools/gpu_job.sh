#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
Q="--no-cpu-baseline --no-kernel-timing --no-e2e --no-parity --no-subrecords"
for v in 1 0 1 0; do
DTS_XP_F32_TILE192=$v timeout -k 10 300 python bench.py --steps 4 --warmup 1 --dtype f32 $Q 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 tile192=$v', 'ms/step', d['ms_per_step'], 'evals/s', d['value'])" || exit 1
done
timeout -k 10 600 python -m pytest tests/test_gpu_search.py tests/test_gpu_ops.py -m gpu -q --timeout 300 -k "search_parity and float32 or test_conv2d" 2>&1 | tail -2
