#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2aa; mkdir -p $O
timeout -k 10 400 python tools/conv_bench.py --n 64 --no-res --stats --variants conv_variant=1 conv_variant=61 conv_variant=71 > $O/order.txt 2>&1; grep -v amdgpu $O/order.txt | grep "3x3\|totals" | cut -c1-330
