#!/bin/bash
# Stall attribution of the Newton kernels: rocprofv3 --pmc passes (counters only, one group of <= 8 SQ counters per pass) over
# tools/stall_child.py at several occupancies.  Usage (GPU box, via gpurun): tools/stall_pmc.sh <tag> [cfg] ["B list"]; output gpurun_out/stall_<tag>/B<B>/<group>/...
TAG=${1:-r6}
CFG=${2:-3}
BS=${3:-"1 256 1024 4096"}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/stall_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
declare -A G
G[time]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS GRBM_GUI_ACTIVE"
G[active]="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM"
G[insts]="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_FLAT"
G[ifetch]="SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES"
G[scalar]="SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL SQC_TC_DATA_READ_REQ SQC_ICACHE_INPUT_VALID_READYB SQC_DCACHE_INPUT_VALID_READYB SQC_DCACHE_BUSY_CYCLES"
G[lds]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_THREAD_CYCLES_VALU"
G[vmem]="SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64"
G[l2]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
for B in $BS; do
  mkdir -p $OUT/B$B
  python $R/tools/stall_child.py $CFG $B 3 > $OUT/B$B/plain.txt 2>&1   # un-profiled: the launch times the counters are read against
  for g in ${STALL_GROUPS:-time active insts ifetch scalar lds vmem l2}; do
    rocprofv3 --output-format csv --pmc ${G[$g]} -d $OUT/B$B/$g -o pmc -- python $R/tools/stall_child.py $CFG $B 3 > $OUT/B$B/$g.txt 2> $OUT/B$B/$g.log || echo "pass $g at B $B failed" >> $OUT/failed.txt
    # keep the merge small: only the Newton / solve kernels' rows
    for f in $(find $OUT/B$B/$g -name "*counter_collection.csv"); do
      python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("newton_kernel", "solve_kernel_fast", "nw_sort"))]
w = csv.DictWriter(open(sys.argv[1], "w"), fieldnames=["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value"], extrasaction="ignore")
w.writeheader()
for r in keep:
    r["Kernel_Name"] = r["Kernel_Name"].split("(")[0]
    w.writerow(r)
PY
    done
    rm -f $(find $OUT/B$B/$g -name "*agent_info.csv")
  done
done
python $R/tools/stall_summary.py $OUT $OUT/stall_attribution.json
