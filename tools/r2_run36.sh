#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2aj; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -q -x --timeout 800 -k "config2 or config1" --durations=5 > $O/pytest.log 2>&1; echo "rc=$?"; tail -12 $O/pytest.log
