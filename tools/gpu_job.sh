#!/bin/bash
# scratch GPU job of the moment (one gpurun call): edit, run, read gpurun_out/job_*.  Committed form = the last job run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/ -m gpu -q --timeout 500 > $O/job_suite.log 2>&1 || { tail -30 $O/job_suite.log; exit 1; }
tail -1 $O/job_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v.get('value') for k,v in d['sub_records'].items()})"
