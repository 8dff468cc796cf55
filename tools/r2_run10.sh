#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2j; mkdir -p $O
timeout -k 10 900 python -m pytest tests/ -m gpu -q -x --timeout 800 --deselect tests/test_gpu_fullsize.py > $O/pytest.log 2>&1; echo "rc=$?"; tail -6 $O/pytest.log
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --conv-table > $O/bench64.json 2> $O/bench64.err && cat $O/bench64.json
DTS_GN_FUSE=0 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > $O/bench64_nofuse.json 2> $O/bench64_nofuse.err && cat $O/bench64_nofuse.json
timeout -k 10 200 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-e2e --candidates 8 > $O/bench8.json 2> $O/bench8.err && cat $O/bench8.json
timeout -k 10 200 python tools/vae_bench.py --n 16 > $O/vae.txt 2>&1; cat $O/vae.txt
