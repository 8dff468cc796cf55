#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2h; mkdir -p $O
timeout -k 10 900 python -m pytest tests/ -m gpu -q -x --timeout 800 --deselect tests/test_gpu_fullsize.py > $O/pytest.log 2>&1; echo "rc=$?"; tail -5 $O/pytest.log
timeout -k 10 300 python tools/conv_bench.py --n 64 --stats --iters 7 --variants conv_variant=0 conv_variant=-1 > $O/conv_ab_res.txt 2>&1; cat $O/conv_ab_res.txt
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --conv-table > $O/bench64.json 2> $O/bench64.err && cat $O/bench64.json
DTS_CONV_VARIANT=0 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > $O/bench64_v1.json 2> $O/bench64_v1.err && cat $O/bench64_v1.json
timeout -k 10 200 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-e2e --candidates 8 > $O/bench8.json 2> $O/bench8.err && cat $O/bench8.json
