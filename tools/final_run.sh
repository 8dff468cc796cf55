#!/bin/bash
# The round's evidence run on the GPU box: GPU tests, the bench line (with CPU baseline), rocprofv3 kernel statistics of the same
# command, and the two PMC passes for HBM traffic.  Outputs land in gpurun_out/final_*; copy the summaries into profiles/.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/ -m gpu -q --timeout 500 > $O/final_pytest.log 2>&1 || { tail -20 $O/final_pytest.log; exit 1; }
tail -2 $O/final_pytest.log
python bench.py --steps 8 --warmup 2 > $O/final_bench.json 2> $O/final_bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/final_prof.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/final_pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/final_pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/final_pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/final_pmc_write.log 2>&1 || exit 1
cat $O/final_bench.json
