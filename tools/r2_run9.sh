#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2i; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_vae.py tests/test_gpu_sd_shapes.py tests/test_gpu_ops.py -m gpu -q -x -s --timeout 800 > $O/pytest.log 2>&1; echo "rc=$?"; grep -vE "^\s*$|transformers\]" $O/pytest.log | tail -25
