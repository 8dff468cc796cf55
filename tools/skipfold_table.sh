#!/bin/bash
# per-shape conv tables with and without the skip fold (stderr of bench.py --conv-table), one box
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
Q="--no-cpu-baseline --no-e2e --no-parity --no-subrecords"
for v in 1 0 ${SF_EXTRA}; do
  DTS_CONV_SKIP_FOLD=$v python bench.py --steps 6 --warmup 2 $Q --conv-table > $O/sft_bench_$v.json 2> $O/sft_bench_$v.err || { tail -5 $O/sft_bench_$v.err; exit 1; }
  python -c "
import json,sys; d=json.loads(open('$O/sft_bench_$v.json').read().strip().splitlines()[-1]); print('fold=$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernels'))"
done
