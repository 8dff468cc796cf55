#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2v; mkdir -p $O
timeout -k 10 400 python tools/conv_bench.py --n 64 --stats --variants conv_variant=0 conv_variant=0,conv_splits=2 conv_variant=0,conv_splits=3 conv_variant=0,conv_tile=128 conv_variant=0,conv_tile=128,conv_splits=2 conv_variant=0,conv_tile=128,conv_splits=3 > $O/l3.txt 2>&1
grep "L3\|L2" $O/l3.txt | cut -c1-400
