#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2n; mkdir -p $O
timeout -k 10 900 python -m pytest tests/ -m gpu -q -x --timeout 800 --deselect tests/test_gpu_fullsize.py > $O/pytest.log 2>&1; rc=$?; echo "rc=$rc"; tail -6 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > $O/bench64.json 2> $O/bench64.err && cat $O/bench64.json
DTS_CONV_FIXUP=0 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > $O/bench64_nofix.json 2> $O/bench64_nofix.err && cat $O/bench64_nofix.json
timeout -k 10 200 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-e2e --candidates 8 > $O/bench8.json 2> $O/bench8.err && cat $O/bench8.json
DTS_CONV_FIXUP=0 timeout -k 10 200 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-e2e --candidates 8 > $O/bench8_nofix.json 2> $O/bench8_nofix.err && cat $O/bench8_nofix.json
