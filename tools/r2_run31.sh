#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ae; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_vae.py -q -x --timeout 500 > $O/pytest.log 2>&1; echo "rc=$?"; tail -6 $O/pytest.log
timeout -k 10 300 python tools/sd_bench.py > $O/sd.txt 2>&1; grep -v amdgpu $O/sd.txt | tail -3
timeout -k 10 300 python tools/sd_bench.py --scorer brightness > $O/sd_b.txt 2>&1; grep -v amdgpu $O/sd_b.txt | tail -3
