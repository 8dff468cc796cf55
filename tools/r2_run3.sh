#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_bench_cli.py tests/test_gpu_sharded.py -m gpu -q -x -s --timeout 800 > $O/pytest_new.log 2>&1; echo "rc=$?"; grep -E "config|passed|failed|decisions|Error" $O/pytest_new.log | tail -30
