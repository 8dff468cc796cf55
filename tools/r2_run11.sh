#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2k; mkdir -p $O
timeout -k 10 900 python -m pytest tests/ -m gpu -q -x --timeout 800 --deselect tests/test_gpu_fullsize.py > $O/pytest.log 2>&1; echo "rc=$?"; tail -6 $O/pytest.log
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > $O/bench64.json 2> $O/bench64.err && cat $O/bench64.json
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --dtype f16 > $O/bench64_f16.json 2> $O/bench64_f16.err && cat $O/bench64_f16.json
timeout -k 10 200 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-e2e --candidates 8 > $O/bench8.json 2> $O/bench8.err && cat $O/bench8.json
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof8 -o runc -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-e2e --no-parity --candidates 8 > $O/prof8.log 2>&1; echo "prof rc=$?"
