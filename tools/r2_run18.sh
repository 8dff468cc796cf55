#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2r; mkdir -p $O
timeout -k 10 900 python -m pytest tests/ -m gpu -q -x --timeout 800 --deselect tests/test_gpu_fullsize.py > $O/pytest.log 2>&1; rc=$?; echo "rc=$rc"; tail -8 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/conv_bench.py --set 128 --n 4 --no-res --stats --variants conv_variant=0 conv_variant=1 > $O/conv128_n4.txt 2>&1; grep -v amdgpu $O/conv128_n4.txt
timeout -k 10 300 python tools/conv_bench.py --set 128 --n 4 --stats --variants conv_variant=0 conv_variant=1 > $O/conv128_n4_res.txt 2>&1; grep -v amdgpu $O/conv128_n4_res.txt
timeout -k 10 200 python tools/vae_bench.py --n 16 > $O/vae.txt 2>&1; grep -v amdgpu $O/vae.txt
DTS_CONV_VARIANT=0 timeout -k 10 200 python tools/vae_bench.py --n 16 > $O/vae_v0.txt 2>&1; grep -v amdgpu $O/vae_v0.txt
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --conv-table > $O/bench64.json 2> $O/bench64.err && cut -c1-1200 $O/bench64.json
