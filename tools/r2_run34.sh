#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ah; mkdir -p $O
timeout -k 10 300 python tools/conv_bench.py --n 64 --no-res --stats --variants conv_variant=1 conv_variant=91 > $O/persist.txt 2>&1; grep -v amdgpu $O/persist.txt | grep "3x3\|totals" | cut -c1-260
timeout -k 10 300 python tools/conv_bench.py --n 64 --stats --variants conv_variant=1 conv_variant=91 > $O/persist_res.txt 2>&1; grep -v amdgpu $O/persist_res.txt | grep "3x3\|totals" | cut -c1-260
PP_STRESS_VARIANT=91 timeout -k 10 400 python tools/pp_stress.py > $O/stress91.txt 2>&1; echo "stress rc=$?"; grep -c " ok" $O/stress91.txt; grep "FAIL" $O/stress91.txt | head
