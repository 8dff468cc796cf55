#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2o; mkdir -p $O
timeout -k 10 400 python tools/conv_bench.py --n 128 --variants conv_epi=0 conv_epi=1 conv_epi=2 conv_epi=3 > $O/conv_epi.txt 2>&1; cat $O/conv_epi.txt
for e in 0 1 2 3; do DTS_CONV_EPI=$e timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-kernel-timing > $O/bench_epi$e.json 2> $O/bench_epi$e.err; python -c "import json,sys;d=json.loads(open('$O/bench_epi$e.json').read().strip().splitlines()[-1]);print('epi',$e,d['value'],d['ms_per_step'])"; done
