#!/bin/bash
# rocprofv3 recipe for the bench (run on the GPU box through gpurun): kernel-trace stats in one run, PMC
# counters in their own runs (never combined with trace domains), outputs under gpurun_out/prof_<tag>/.
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --cpu-sample 0 --no-stages --no-configs --no-parity --no-scaling-preview --streams 0 --no-live-traffic --details ''"  # the timed pattern of the default command only: 3 warm-ups + 10 single-batch solves on one stream (+ the 11 solves of the order-hint leg: same kernel)
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- $BENCH > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- $BENCH > /dev/null 2> $OUT/pmc_write.log
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d $OUT/pmc_sq -o sq -- $BENCH > /dev/null 2> $OUT/pmc_sq.log
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 -d $OUT/pmc_f64 -o f64 -- $BENCH > /dev/null 2> $OUT/pmc_f64.log
find $OUT -name "*.csv" | head -30
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f; done
for f in $(find $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_f64 -name "*counter_collection.csv"); do echo "== $f"; head -3 $f; python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "")[:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
for k, d in agg.items():
    print(k, dict(d))
PY
done
cat $OUT/bench_trace.json | tail -1
