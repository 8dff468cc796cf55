#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ad; mkdir -p $O
for v in 0 1 0 1; do DTS_GN_PREFER_FUSED=$v timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-kernel-timing > $O/b64_$v.json 2> $O/b64_$v.err; python -c "import json;d=json.loads(open('$O/b64_$v.json').read().strip().splitlines()[-1]);print('N=64 prefer_fused',$v,d['value'],d['ms_per_step'])"; done
for v in 0 1 0 1; do DTS_GN_PREFER_FUSED=$v timeout -k 10 300 python bench.py --steps 20 --warmup 2 --candidates 8 --no-cpu-baseline --no-e2e --no-kernel-timing > $O/b8_$v.json 2> $O/b8_$v.err; python -c "import json;d=json.loads(open('$O/b8_$v.json').read().strip().splitlines()[-1]);print('n=8 prefer_fused',$v,d['value'],d['ms_per_step'])"; done
