set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-kernel-timing --no-e2e --no-parity --no-subrecords"
rm -rf $O/r6_prof_x3
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r6_prof_x3 -- python3 $R/bench.py --steps 3 --warmup 1 $Q > $O/r6_prof_x3.log 2>&1 || exit 1
find $O/r6_prof_x3 -name "*kernel_trace.csv" -delete; find $O/r6_prof_x3 -name "*.db" -delete
head -14 $O/r6_prof_x3/*/*kernel_stats.csv | cut -c1-200
