#!/bin/bash
# same-box A/B of two builds of the library: tools/ab/libdts_old.so (DTS_LIB_PATH) vs the in-tree one
for i in 1 2; do
  DTS_LIB_PATH=$PWD/tools/ab/libdts_old.so timeout -k 10 200 python tools/conv_bench.py $CONV_BENCH_ARGS > gpurun_out/ab_old$i.txt 2>&1 || exit 1
  timeout -k 10 200 python tools/conv_bench.py $CONV_BENCH_ARGS > gpurun_out/ab_new$i.txt 2>&1 || exit 1
done
cd gpurun_out && paste <(grep TFLOP ab_old1.txt | awk '{print $1,$2,$3, $(NF-1)}') <(grep TFLOP ab_new1.txt | awk '{print $(NF-1)}') <(grep TFLOP ab_old2.txt | awk '{print $(NF-1)}') <(grep TFLOP ab_new2.txt | awk '{print $(NF-1)}')
