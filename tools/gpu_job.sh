#!/bin/bash
# scratch GPU job (one gpurun call): the GPU suite, the smoke test and a short bench line.  Edit for the experiment of the moment; read gpurun_out/job_*.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/ -m gpu -q --timeout 900 > $O/job_suite.log 2>&1 || { tail -40 $O/job_suite.log | cut -c1-300; exit 1; }
tail -1 $O/job_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity --no-e2e --no-subrecords 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['whole_step_frac'])"
