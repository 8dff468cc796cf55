#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "ping_pong or repeat_bit" 2>&1 | tail -3 || exit 1
python tools/conv_bench.py --n 64 --stats --variants conv_variant=1 2>&1 | grep -v amdgpu | grep "3x3\|total" | cut -c1-120
DTS_LIB_PATH=$R/diffusion_tts_amd/libdts_hip_old.so python tools/conv_bench.py --n 64 --stats --variants conv_variant=1 2>&1 | grep -v amdgpu | grep "3x3\|total" | cut -c1-120
