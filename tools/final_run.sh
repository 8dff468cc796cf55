#!/bin/bash
# The round's evidence run on the GPU box, part A (one gpurun call, ~15 min): GPU tests, the bench line in the headline mode (f16x3: CPU baseline
# + parity + whole-search index agreement + e2e + sub-records, with bf16 / f16 in other_dtype), the bf16 / f32 lines with their own conv
# rooflines, the secondary workloads.  Part B (tools/final_profiles.sh, a second call) takes the rocprofv3 kernel statistics and the PMC passes.
# Outputs land in gpurun_out/final_*; tools/collect_profiles.sh copies the summaries into profiles/ under this round's names.
# Multi-GPU checklist for the first lease with >= 2 GPUs (the pool gives one GPU per box, so this section only runs where it can):
#   HSA_ENABLE_IPC_MODE_LEGACY=0 exported (RCCL needs dmabuf IPC here), NCCL_DEBUG=VERSION to log the RCCL build, `rccl_ranks == N` and
#   `dist_backend == "nccl"` in every line, N = 1, 2, 4, 8 back to back.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/final_*
DTS_TEST_SHARDED_FULLSIZE=1 timeout -k 10 1100 python -m pytest tests/ -m gpu -q --timeout 900 --durations=12 > $O/final_pytest.log 2>&1 || { tail -20 $O/final_pytest.log; exit 1; }
tail -2 $O/final_pytest.log
python bench.py --steps 10 --warmup 3 --conv-table --conv-sequence $O/final_conv_sequence.json > $O/final_bench.json 2> $O/final_bench.err || { tail -5 $O/final_bench.err; exit 1; }
Q="--no-cpu-baseline --no-e2e --no-parity --no-subrecords"
python bench.py --steps 20 --warmup 5 --dtype bf16 --conv-table --conv-sequence $O/final_conv_sequence_bf16.json $Q > $O/final_bench_bf16.json 2> $O/final_bench_bf16.err || exit 1
python bench.py --steps 4 --warmup 1 --dtype f32 $Q > $O/final_bench_f32.json 2> $O/final_bench_f32.err || exit 1
timeout -k 10 600 python bench.py --workload adm64_mcts --S 256 > $O/final_bench_mcts.json 2> $O/final_bench_mcts.err || exit 1
DTS_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 6 --no-kernel-timing > $O/final_bench_2rank_gloo.json 2> $O/final_bench_2rank_gloo.err || exit 1
DTS_SHARD_ALWAYS_COLLECT=1 NCCL_DEBUG=VERSION python bench.py --steps 6 --warmup 2 $Q --no-kernel-timing > $O/final_bench_rccl1.json 2> $O/final_bench_rccl1.err || exit 1
NG=$(python -c "import torch; print(torch.cuda.device_count())")
if [ "$NG" -ge 2 ]; then
  export HSA_ENABLE_IPC_MODE_LEGACY=0 NCCL_DEBUG=VERSION
  # FIRST on a multi-GPU lease: the RCCL variants of the sharded tests -- tiny nets, and BASELINE configs[2] at full size walked along the reference's
  # own run with one rank per GPU (tests/test_gpu_sharded.py::test_sharded_config3_walk_against_the_reference_run[nccl]: rccl ranks == GPUs, 72 collectives,
  # rewards within 5e-8, every decidable decision equal, final image)
  timeout -k 10 1500 python -m pytest tests/test_gpu_sharded.py -m gpu -q -k nccl -s > $O/final_pytest_rccl.log 2>&1 || { tail -30 $O/final_pytest_rccl.log; exit 1; }
  tail -3 $O/final_pytest_rccl.log
  for n in 1 2 4 8; do
    [ "$n" -le "$NG" ] || continue
    python bench.py --gpus $n --steps 10 --warmup 3 $Q > $O/final_scale_n$n.json 2> $O/final_scale_n$n.err || exit 1
    python - "$O/final_scale_n$n.json" $n <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); n = int(sys.argv[2])
assert d['n_gpus'] == n and (n == 1 or (d['rccl_ranks'] == n and d['dist_backend'] == 'nccl')), d
print('scale', n, d['value'], d.get('weak_value'))
PY
  done
fi
cat $O/final_bench.json | cut -c1-600
