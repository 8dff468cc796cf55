#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ag; mkdir -p $O
timeout -k 10 300 python tools/conv_bench.py --n 64 --stats --variants conv_variant=0 conv_variant=1 conv_variant=0 conv_variant=1 > $O/res.txt 2>&1; grep -v amdgpu $O/res.txt | grep "3x3\|totals" | cut -c1-330
