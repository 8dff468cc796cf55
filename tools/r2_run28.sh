#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ab; mkdir -p $O
timeout -k 10 400 python tools/conv_bench.py --n 64 --no-res --stats --variants conv_variant=1 conv_variant=81 > $O/split.txt 2>&1; grep -v amdgpu $O/split.txt | grep "3x3\|totals" | cut -c1-260
timeout -k 10 500 python tools/pp_stress.py > $O/stress.txt 2>&1; echo "stress rc=$?"; tail -2 $O/stress.txt
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -x --timeout 500 > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-kernel-timing > $O/bench64.json 2> $O/bench64.err; cut -c1-260 $O/bench64.json
DTS_CONV_VARIANT=81 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-kernel-timing > $O/bench64_81.json 2> $O/bench64_81.err; cut -c1-260 $O/bench64_81.json
