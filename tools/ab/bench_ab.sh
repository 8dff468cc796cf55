#!/bin/bash
# same-box A/B of the whole bench step: tools/ab/libdts_old.so (DTS_LIB_PATH) vs the in-tree library
for i in 1 2; do
  DTS_LIB_PATH=$PWD/tools/ab/libdts_old.so python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old', d['value'], d['ms_per_step'], 'conv ms/step', d['roofline']['conv_ms_per_step'])" || exit 1
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new', d['value'], d['ms_per_step'], 'conv ms/step', d['roofline']['conv_ms_per_step'])" || exit 1
done
