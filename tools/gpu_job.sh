#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --timeout 900 -s -k reduced > $O/job_suite.log 2>&1 || { grep -a -v "^split precision" $O/job_suite.log | tail -40; exit 1; }
grep -a "naive sampler, tiny\|config 3 reduced\|decidable selections" $O/job_suite.log; tail -1 $O/job_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 10 --warmup 2 > $O/job_bench.json 2> $O/job_bench.err || { tail -20 $O/job_bench.err; exit 1; }
grep -a "bench +" $O/job_bench.err | grep -v " conv (" | cut -c1-400 | tail -30
