#!/bin/bash
# Dev tool (GPU box): GPU test suite, per-phase shader clocks of the headline kernel, bare iteration cost, instruction-cache / stall counters.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/diag
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $OUT/gpu_tests.log 2>&1
echo "pytest exit $?" >> $OUT/gpu_tests.log
tail -25 $OUT/gpu_tests.log
timeout 300 python tools/phase_cycles.py 4096 3 > $OUT/phase.log 2>&1; tail -4 $OUT/phase.log
timeout 600 python tools/iter_cost.py > $OUT/iter_cost.log 2>&1; cat $OUT/iter_cost.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQC_ICACHE_[A-Z_]+|SQ_WAIT_[A-Z_]+|SQ_INST_CYCLES_[A-Z_]+|SQ_INSTS_[A-Z_0-9]+|SQ_ACTIVE_INST_[A-Z_]+|SQ_IFETCH[A-Z_]*" | sort -u | tr '\n' ' ' > $OUT/counters.txt
cat $OUT/counters.txt
BENCH="python $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --streams 1 --no-stages"
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --output-format csv --pmc $set -d $OUT/pmc_$tag -o p -- $BENCH > /dev/null 2> $OUT/pmc_$tag.log
  f=$(find $OUT/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "")
    if "solve_kernel_fast" not in k: continue
    agg[k[:50]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items(): print(k, dict(d))
PY
  rm -rf $OUT/pmc_$tag/*/*.db 2>/dev/null
done
