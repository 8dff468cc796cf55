#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2s; mkdir -p $O
timeout -k 10 200 python tools/pp1x1_bisect.py > $O/bisect.txt 2>&1; grep -v amdgpu $O/bisect.txt
