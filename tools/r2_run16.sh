#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2p; mkdir -p $O
for e in 0 1 2 3; do DTS_GN_NT=$e timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-kernel-timing > $O/bench_nt$e.json 2> $O/bench_nt$e.err; python -c "import json,sys;d=json.loads(open('$O/bench_nt$e.json').read().strip().splitlines()[-1]);print('nt',$e,d['value'],d['ms_per_step'])"; done
