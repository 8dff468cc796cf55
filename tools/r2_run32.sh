#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2af; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -x --timeout 500 > $O/pytest.log 2>&1; rc=$?; echo "rc=$rc"; tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/conv_bench.py --n 32 --up --stats --variants conv_variant=0 conv_variant=1 > $O/up_adm.txt 2>&1; grep -v amdgpu $O/up_adm.txt | grep "3x3\|totals" | cut -c1-230
timeout -k 10 300 python tools/conv_bench.py --set 128 --n 2 --up --stats --variants conv_variant=0 conv_variant=1 > $O/up_128.txt 2>&1; grep -v amdgpu $O/up_128.txt | grep "3x3\|CLS\|VAE\|totals" | cut -c1-230
timeout -k 10 200 python tools/vae_bench.py --n 16 > $O/vae.txt 2>&1; grep -v amdgpu $O/vae.txt
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-kernel-timing > $O/bench64.json 2> $O/bench64.err; cut -c1-220 $O/bench64.json
