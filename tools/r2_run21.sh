#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2u; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_vae.py -q -x --timeout 500 > $O/pytest.log 2>&1; echo "rc=$?"; tail -5 $O/pytest.log
DTS_SHARD_ALWAYS_COLLECT=1 NCCL_DEBUG=VERSION timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-kernel-timing > $O/bench_rccl1.json 2> $O/bench_rccl1.err; echo "rc=$?"; tail -3 $O/bench_rccl1.err; cut -c1-900 $O/bench_rccl1.json
timeout -k 10 200 python tools/vae_bench.py --n 16 > $O/vae.txt 2>&1; grep -v amdgpu $O/vae.txt
