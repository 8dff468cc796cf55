#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2y; mkdir -p $O
timeout -k 10 600 python tools/dtype_mix.py > $O/mix.txt 2>&1; echo "rc=$?"; grep -v amdgpu $O/mix.txt | tail -14
