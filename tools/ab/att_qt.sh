#!/bin/bash
# same-box comparison of the attention query-tiles-per-wave setting (env switch of the shipped library)
DTS_ATT_QT=2 timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k att 2>&1 | tail -2 || exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k att 2>&1 | tail -2 || exit 1
for i in 1 2; do
  for qt in 1 0; do
    DTS_ATT_QT=$qt python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('DTS_ATT_QT=$qt', d['value'], d['ms_per_step'])" || exit 1
  done
done
