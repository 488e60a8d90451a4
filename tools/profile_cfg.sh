#!/bin/bash
# rocprofv3 evidence for ONE config at the headline setting (VERDICT r5 missing 4: only config 3 had counter files): kernel trace + stats in one run, then PMC
# passes (counters only, never with a trace domain) for HBM bytes, fp64 instruction counts and the resident-time split.  The profiled command is tools/stall_child.py
# with the engine's OWN slicing decision (STALL_NO_FORCE_SLICE=1) — what bench.py's config legs run.
#   tools/profile_cfg.sh <tag> <cfg: 2 | 3 | 5 | 3k> [B]     ->  gpurun_out/prof_<tag>/c<cfg>/...
TAG=${1:-r6}
CFG=${2:-5}
B=${3:-4096}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG/c$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export STALL_NO_FORCE_SLICE=1
CMD="python $R/tools/stall_child.py $CFG $B 5"
timeout 300 $CMD > $OUT/plain.txt 2>&1
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.txt 2> $OUT/trace.log
for g in "fetch FETCH_SIZE" "write WRITE_SIZE" "f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_INSTS" \
         "time SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  set -- $g; name=$1; shift
  timeout 300 rocprofv3 --output-format csv --pmc $@ -d $OUT/pmc_$name -o pmc -- $CMD > /dev/null 2> $OUT/pmc_$name.log || echo "pass $name failed" >> $OUT/failed.txt
done
rm -f $(find $OUT -name "*agent_info.csv") $(find $OUT/trace -name "*kernel_trace.csv")
python $R/tools/profile_cfg_summary.py $OUT $CFG $B
