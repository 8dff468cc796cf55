#!/bin/bash
# The round's evidence run on the GPU box: GPU tests, the bench line (CPU baseline + parity + e2e), rocprofv3 kernel statistics of the same
# command, the two PMC passes for HBM traffic, and the secondary workloads.  Outputs land in gpurun_out/final_*; copy the summaries into profiles/.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; rm -rf $O/final_*
timeout -k 10 900 python -m pytest tests/ -m gpu -q --timeout 800 > $O/final_pytest.log 2>&1 || { tail -20 $O/final_pytest.log; exit 1; }
tail -2 $O/final_pytest.log
python bench.py --steps 8 --warmup 2 --conv-table --conv-sequence $O/final_conv_sequence.json > $O/final_bench.json 2> $O/final_bench.err || exit 1
python bench.py --workload ddpmpp32_rejection --conv-table > $O/final_bench_rej32.json 2> $O/final_bench_rej32.err || exit 1
python bench.py --steps 20 --warmup 2 --candidates 8 --no-cpu-baseline --no-e2e > $O/final_bench_cand8.json 2> $O/final_bench_cand8.err || exit 1
timeout -k 10 600 python bench.py --workload adm64_mcts --S 256 > $O/final_bench_mcts.json 2> $O/final_bench_mcts.err || exit 1
DTS_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 4 --no-kernel-timing > $O/final_bench_2rank_gloo.json 2> $O/final_bench_2rank_gloo.err || exit 1
python tools/vae_bench.py --n 16 > $O/final_vae.txt 2>&1 || exit 1
DTS_SHARD_ALWAYS_COLLECT=1 NCCL_DEBUG=VERSION python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-kernel-timing > $O/final_bench_rccl1.json 2> $O/final_bench_rccl1.err || exit 1
python tools/pp_stress.py > $O/final_pp_stress.txt 2>&1 || exit 1
python tools/sd_bench.py --scorer brightness > $O/final_sd.txt 2>&1 || exit 1
python tools/sd_bench.py >> $O/final_sd.txt 2>&1 || exit 1
python tools/att_bench.py --n 64 --variants att_xcd=0 att_xcd=1 > $O/final_att.txt 2>&1 || exit 1
python bench.py --steps 10 --warmup 2 --dtype f16 --no-cpu-baseline --no-e2e > $O/final_bench_f16.json 2> $O/final_bench_f16.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-e2e > $O/final_prof.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/final_pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-e2e > $O/final_pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/final_pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-e2e > $O/final_pmc_write.log 2>&1 || exit 1
cd $R && python tools/pmc_traffic.py $O/final_pmc_fetch $O/final_pmc_write $O/final_hbm_traffic_pmc.json $O/final_conv_sequence.json > $O/final_pmc_per_shape.txt 2>&1
find $O -name "*kernel_trace.csv" -size +20M -delete; find $O -name "*.db" -delete
cat $O/final_bench.json
