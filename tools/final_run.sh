#!/bin/bash
# The round's evidence run on the GPU box (one gpurun call, ~12 min): GPU tests, the bench line (CPU baseline + parity + whole-search index
# agreement + e2e + sub-records), rocprofv3 kernel statistics of the same command, the two PMC passes for HBM traffic, and the secondary
# workloads.  Outputs land in gpurun_out/final_*; tools/collect_profiles.sh copies the summaries into profiles/ under this round's names.
# Multi-GPU checklist for the first lease with >= 2 GPUs (the pool gives one GPU per box, so this section only runs where it can):
#   HSA_ENABLE_IPC_MODE_LEGACY=0 exported (RCCL needs dmabuf IPC here), NCCL_DEBUG=VERSION to log the RCCL build, `rccl_ranks == N` and
#   `dist_backend == "nccl"` in every line, N = 1, 2, 4, 8 back to back.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/final_*
timeout -k 10 1100 python -m pytest tests/ -m gpu -q --timeout 900 > $O/final_pytest.log 2>&1 || { tail -20 $O/final_pytest.log; exit 1; }
tail -2 $O/final_pytest.log
python bench.py --steps 20 --warmup 5 --conv-table --conv-sequence $O/final_conv_sequence.json > $O/final_bench.json 2> $O/final_bench.err || { tail -5 $O/final_bench.err; exit 1; }
python bench.py --steps 20 --warmup 5 --dtype f16 --no-cpu-baseline --no-e2e --no-parity --no-subrecords > $O/final_bench_f16.json 2> $O/final_bench_f16.err || exit 1
# the two modes that meet the north star's tolerance, each with its own conv roofline (f32: v_mfma_f32_16x16x4_f32; f16x3: split precision on the 16-bit MFMA)
python bench.py --steps 10 --warmup 2 --dtype f16x3 --conv-table --no-cpu-baseline --no-e2e --no-parity --no-subrecords > $O/final_bench_f16x3.json 2> $O/final_bench_f16x3.err || exit 1
python bench.py --steps 4 --warmup 1 --dtype f32 --no-cpu-baseline --no-e2e --no-parity --no-subrecords > $O/final_bench_f32.json 2> $O/final_bench_f32.err || exit 1
timeout -k 10 600 python bench.py --workload adm64_mcts --S 256 > $O/final_bench_mcts.json 2> $O/final_bench_mcts.err || exit 1
DTS_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 6 --no-kernel-timing > $O/final_bench_2rank_gloo.json 2> $O/final_bench_2rank_gloo.err || exit 1
DTS_SHARD_ALWAYS_COLLECT=1 NCCL_DEBUG=VERSION python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-parity --no-subrecords --no-kernel-timing > $O/final_bench_rccl1.json 2> $O/final_bench_rccl1.err || exit 1
NG=$(python -c "import torch; print(torch.cuda.device_count())")
if [ "$NG" -ge 2 ]; then
  export HSA_ENABLE_IPC_MODE_LEGACY=0 NCCL_DEBUG=VERSION
  for n in 1 2 4 8; do
    [ "$n" -le "$NG" ] || continue
    python bench.py --gpus $n --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-parity --no-subrecords > $O/final_scale_n$n.json 2> $O/final_scale_n$n.err || exit 1
    python - "$O/final_scale_n$n.json" $n <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); n = int(sys.argv[2])
assert d['n_gpus'] == n and (n == 1 or (d['rccl_ranks'] == n and d['dist_backend'] == 'nccl')), d
print('scale', n, d['value'], d.get('weak_value'))
PY
  done
fi
python tools/vae_bench.py --n 16 > $O/final_vae.txt 2>&1 || exit 1
{ python tools/sd_bench.py --scorer brightness; python tools/sd_bench.py --host-preprocess; python tools/sd_bench.py; DTS_DIST_BACKEND=gloo python tools/sd_bench.py --gpus 2; } > $O/final_sd.txt 2>&1 || exit 1
python tools/att_bench.py --n 64 > $O/final_att.txt 2>&1 || exit 1
# (the conv kernels' main loops are unchanged since round 3: profiles/r03_conv_variants.txt and r03_conv_stamps.txt still describe them; tools/conv_stamps.py
#  rebuilds its -DDTS_STAMPS library on demand)
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-kernel-timing --no-e2e --no-parity --no-subrecords"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_prof -- python3 $R/bench.py --steps 3 --warmup 1 $Q > $O/final_prof.log 2>&1 || exit 1
# the SAME process measured both ways: bench.py's in-process per-dispatch HIP events (roofline.avg_launch_us in its JSON line) and rocprofv3's trace of those dispatches
rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_prof_both -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-parity --no-subrecords > $O/final_prof_both.json 2> $O/final_prof_both.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_prof_c8 -- python3 $R/bench.py --steps 6 --warmup 1 --candidates 8 $Q > $O/final_prof_c8.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_prof_x3 -- python3 $R/bench.py --steps 3 --warmup 1 --dtype f16x3 $Q > $O/final_prof_x3.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/final_pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 $Q > $O/final_pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/final_pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 $Q > $O/final_pmc_write.log 2>&1 || exit 1
cd $R && python tools/pmc_traffic.py $O/final_pmc_fetch $O/final_pmc_write $O/final_hbm_traffic_pmc.json $O/final_conv_sequence.json > $O/final_pmc_per_shape.txt 2>&1
find $O -name "*kernel_trace.csv" -size +20M -delete; find $O -name "*counter_collection.csv" -size +20M -delete; find $O -name "*.db" -delete
cat $O/final_bench.json | cut -c1-600
