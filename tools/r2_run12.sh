#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2l; mkdir -p $O
timeout -k 10 900 python -m pytest tests/ -m gpu -q -x --timeout 800 --deselect tests/test_gpu_fullsize.py > $O/pytest.log 2>&1; echo "rc=$?"; tail -6 $O/pytest.log
timeout -k 10 300 python tools/att_bench.py --n 64 --variants att_db=0 att_db=1 > $O/att_n64.txt 2>&1; cat $O/att_n64.txt
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > $O/bench64.json 2> $O/bench64.err && cat $O/bench64.json
