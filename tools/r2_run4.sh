#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2d; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_sd_shapes.py tests/test_gpu_sd.py -m gpu -q -s --timeout 800 > $O/pytest_sd.log 2>&1; echo "rc=$?"; grep -vE "^\s*$|transformers\]" $O/pytest_sd.log | tail -40
