#!/bin/bash
# SQ counters of one search iteration per compute mode (two --pmc passes each, kernel trace only; never combined with other trace domains)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-e2e --no-parity --no-subrecords"
for d in bf16 f16x3; do
  rm -rf $O/sq_${d}_a $O/sq_${d}_b
  rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $O/sq_${d}_a -- python3 $R/bench.py --dtype $d $Q > $O/sq_${d}_a.log 2>&1 || { tail -5 $O/sq_${d}_a.log; exit 1; }
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD --output-format csv -d $O/sq_${d}_b -- python3 $R/bench.py --dtype $d $Q > $O/sq_${d}_b.log 2>&1 || { tail -5 $O/sq_${d}_b.log; exit 1; }
  echo "== $d" | tee -a $O/sq_summary.txt
  python3 $R/tools/sq_counters.py $O/sq_${d}_a $O/sq_${d}_b | tee -a $O/sq_summary.txt
  find $O/sq_${d}_a $O/sq_${d}_b -name "*kernel_trace.csv" -delete; find $O/sq_${d}_a $O/sq_${d}_b -name "*.db" -delete; find $O/sq_${d}_a $O/sq_${d}_b -name "*counter_collection.csv" -size +20M -delete
done
