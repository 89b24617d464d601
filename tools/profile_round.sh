#!/bin/bash
# rocprofv3 evidence for the bench workload (run on the GPU box from the repo root):
#   kernel-trace + stats of the default bench command, then SEPARATE --pmc passes (SQ issue counters, instruction cache,
#   FETCH_SIZE, WRITE_SIZE) of a shorter run of the same command.  Everything lands under gpurun_out/prof_round/; the
#   summary tools/summarise_profile.py writes is what gets copied to profiles/.
set -e
OUT=${1:-gpurun_out/prof_round}
ENVARG=${2:-Env03-v2}
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
python3 bench.py --env $ENVARG --steps 300 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/trace -o trace -- python3 $ROOT/bench.py --env $ENVARG --steps 300 --warmup 20 --no-cpu-baseline > $ROOT/$OUT/bench_traced.json 2> $ROOT/$OUT/trace.err
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d $ROOT/$OUT/pmc_$tag -o pmc -- python3 $ROOT/bench.py --env $ENVARG --steps 40 --warmup 10 --no-cpu-baseline > /dev/null 2> $ROOT/$OUT/pmc_$tag.err || echo "pmc pass $tag failed" >> $ROOT/$OUT/errors.log
done
cd $ROOT
python3 tools/summarise_profile.py $OUT $ENVARG > $OUT/summary.json
# the rocpd databases are tens of MB each (gpurun merges at most 64 MiB back): keep the summary and the kernel-trace table only
python3 - "$OUT" <<'PY'
import sqlite3, sys, csv, os
out = sys.argv[1]
con = sqlite3.connect(os.path.join(out, "trace", "trace_results.db"))
with open(os.path.join(out, "kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
    for r in con.execute("select * from top_kernels"):
        w.writerow([r[0][:120], r[1], r[2], r[3], r[4]])
PY
rm -rf $OUT/trace $OUT/pmc_*/
cat $OUT/summary.json
