#!/bin/bash
# A/B on one box: the 1x1 skip convolutions folded into conv1's launch (default) against their own launches (DTS_CONV_SKIP_FOLD=0), alternating.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
Q="--no-cpu-baseline --no-e2e --no-parity --no-subrecords --no-kernel-timing"
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "folds or split_precision" -s > $O/sf_pytest.log 2>&1 || { tail -30 $O/sf_pytest.log; exit 1; }
grep "skip fold\|passed\|failed" $O/sf_pytest.log
for i in 1 2; do
  for v in 1 0; do
    DTS_CONV_SKIP_FOLD=$v python bench.py --steps 10 --warmup 3 $Q > $O/sf_bench_${v}_$i.json 2> $O/sf_bench_${v}_$i.err || { tail -5 $O/sf_bench_${v}_$i.err; exit 1; }
    python -c "
import json,sys; d=json.loads(open('$O/sf_bench_${v}_$i.json').read().strip().splitlines()[-1]); print('fold=$v', d['value'], d['ms_per_step'])"
  done
done
