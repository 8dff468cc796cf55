#!/bin/bash
# The round's evidence run, part B (a second gpurun call, ~12 min): rocprofv3 kernel statistics of the bench command in the headline mode
# (f16x3), in bf16 and at 8 candidates per GPU, the SAME process measured both ways (in-process dispatch events vs the trace), and the two
# PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, --kernel-trace only) for HBM traffic in f16x3 and in bf16.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python tools/vae_bench.py --n 16 > $O/final_vae.txt 2>&1 || exit 1
{ python tools/sd_bench.py --scorer brightness; python tools/sd_bench.py; } > $O/final_sd.txt 2>&1 || exit 1
python tools/att_bench.py --n 64 > $O/final_att.txt 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-kernel-timing --no-e2e --no-parity --no-subrecords"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_prof_x3 -- python3 $R/bench.py --steps 3 --warmup 1 $Q > $O/final_prof_x3.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_prof_bf16 -- python3 $R/bench.py --steps 3 --warmup 1 --dtype bf16 $Q > $O/final_prof_bf16.log 2>&1 || exit 1
# the SAME process measured both ways: bench.py's in-process per-dispatch HIP events (roofline.avg_launch_us in its JSON line) and rocprofv3's trace of those dispatches
rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_prof_both -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-parity --no-subrecords > $O/final_prof_both.json 2> $O/final_prof_both.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/final_prof_c8 -- python3 $R/bench.py --steps 6 --warmup 1 --candidates 8 $Q > $O/final_prof_c8.log 2>&1 || exit 1
for dt in f16x3 bf16; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/final_pmc_fetch_$dt -- python3 $R/bench.py --steps 2 --warmup 1 --dtype $dt $Q > $O/final_pmc_fetch_$dt.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/final_pmc_write_$dt -- python3 $R/bench.py --steps 2 --warmup 1 --dtype $dt $Q > $O/final_pmc_write_$dt.log 2>&1 || exit 1
done
cd $R
# (the per-shape join with part A's gpurun_out/final_conv_sequence*.json happens in tools/collect_profiles.sh, in the build container: gpurun_out/ does
#  not travel to the GPU box, the counter CSVs come back)
find $O -name "*kernel_trace.csv" -size +20M -delete; find $O -name "*counter_collection.csv" -size +30M -delete; find $O -name "*.db" -delete
ls $O/final_prof_x3/*/ $O/final_pmc_fetch_f16x3/*/ | head
