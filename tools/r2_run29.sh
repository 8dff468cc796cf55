#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ac; mkdir -p $O
timeout -k 10 300 python tools/conv_bench.py --n 8 --no-res --stats --variants conv_variant=0 conv_variant=1 > $O/n8.txt 2>&1; grep -v amdgpu $O/n8.txt | grep "3x3\|totals" | cut -c1-230
timeout -k 10 300 python tools/conv_bench.py --n 8 --stats --variants conv_variant=0 conv_variant=1 > $O/n8res.txt 2>&1; grep -v amdgpu $O/n8res.txt | grep "3x3\|totals" | cut -c1-230
timeout -k 10 300 python tools/conv_bench.py --n 16 --no-res --stats --variants conv_variant=0 conv_variant=1 > $O/n16.txt 2>&1; grep -v amdgpu $O/n16.txt | grep "3x3\|totals" | cut -c1-230
