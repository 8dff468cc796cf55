#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2g; mkdir -p $O
DTS_CONV_VARIANT=1 timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout 500 > $O/pytest_pp.log 2>&1; echo "rc=$?"; tail -4 $O/pytest_pp.log
timeout -k 10 400 python tools/conv_bench.py --n 64 --stats --no-res --iters 7 --variants conv_variant=0 conv_variant=1 conv_variant=11 conv_variant=41 conv_variant=51 > $O/conv_diag3.txt 2>&1; cat $O/conv_diag3.txt
