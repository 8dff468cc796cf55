#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2m; mkdir -p $O
timeout -k 10 300 python tools/conv_resid_check.py > $O/resid.txt 2>&1; cat $O/resid.txt
