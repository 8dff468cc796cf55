#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ai; mkdir -p $O
timeout -k 10 900 python -m pytest tests/ -m gpu -q -x --timeout 800 > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?"; cut -c1-240 $O/bench.json
