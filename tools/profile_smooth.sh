#!/bin/bash
# rocprofv3 kernel-trace stats of the reference-smoothing QP engine (tools/smooth_bench.py: 4096 QPs per kind at eps 1e-3 and 1e-4) -> gpurun_out/prof_smooth_<tag>/
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_smooth_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- python $R/tools/smooth_bench.py > $OUT/smooth_bench.log 2> $OUT/trace.log
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats_smoothing.csv; head -12 $f; done
grep eps $OUT/smooth_bench.log
