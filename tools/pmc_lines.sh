#!/bin/bash
# PMC passes over the fused line smoothers of one 128^3 VTI level (through gpurun):
#   bash tools/pmc_lines.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-30)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcl/$tag -o run -- python $R/tools/microbench.py lines --n 128 --case VTI > $R/gpurun_out/pmcl_$tag.log 2>&1
  f=$(ls $R/gpurun_out/pmcl/$tag/*counter_collection.csv 2>/dev/null | head -1)
  echo "== $set"; if [ -n "$f" ]; then python $R/tools/pmc_summary.py $f | grep -i "line_colour" | cut -c1-52,71-140; else tail -3 $R/gpurun_out/pmcl_$tag.log; fi
done
