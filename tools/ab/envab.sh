#!/bin/bash
# same-box A/B of an environment switch of the shipped library: tools/ab/envab.sh VAR OFF ON
V=$1; A=$2; B=$3
for i in 1 2; do
  for val in $A $B; do
    env $V=$val python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V=$val', d['value'], d['ms_per_step'], 'conv ms/step', d['roofline']['conv_ms_per_step'])" || exit 1
  done
done
