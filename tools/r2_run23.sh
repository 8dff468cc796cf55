#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2w; mkdir -p $O
timeout -k 10 500 python tools/pp_stress.py > $O/stress.txt 2>&1; echo "rc=$?"; grep -v amdgpu $O/stress.txt | tail -40
