#!/bin/bash
# round 2, GPU call 1: regression tests after the attention remap + baseline measurements
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2a; mkdir -p $O
timeout -k 10 700 python -m pytest tests/ -m gpu -q -x --timeout 500 > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 200 python tools/att_bench.py --n 64 > $O/att_n64.txt 2>&1 && cat $O/att_n64.txt &&
timeout -k 10 200 python tools/att_bench.py --n 8 > $O/att_n8.txt 2>&1 && cat $O/att_n8.txt &&
timeout -k 10 200 python tools/conv_bench.py --n 64 --stats > $O/conv_n64.txt 2>&1 && cat $O/conv_n64.txt &&
timeout -k 10 200 python tools/conv_bench.py --n 8 --stats > $O/conv_n8.txt 2>&1 && cat $O/conv_n8.txt &&
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --conv-table > $O/bench64.json 2> $O/bench64.err && cat $O/bench64.json &&
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --candidates 32 > $O/bench32.json 2> $O/bench32.err && cat $O/bench32.json &&
timeout -k 10 200 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-e2e --candidates 8 --conv-table > $O/bench8.json 2> $O/bench8.err && cat $O/bench8.json &&
timeout -k 10 300 python bench.py --workload ddpmpp32_rejection --conv-table > $O/bench_rej32.json 2> $O/bench_rej32.err && cat $O/bench_rej32.json &&
DTS_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 4 --no-kernel-timing > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; cat $O/bench_2rank_gloo.json; tail -5 $O/bench_2rank_gloo.err
