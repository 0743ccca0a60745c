#!/bin/bash
# PMC passes over the tiled point smoother at 256^3 (through gpurun, from the repo root):
#   bash tools/pmc_point.sh   -> per-kernel counters, bytes per node in the last column
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-30)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcp/$tag -o run -- python $R/tools/microbench.py point --n 256 --slabs "" > $R/gpurun_out/pmcp_$tag.log 2>&1
  f=$(ls $R/gpurun_out/pmcp/$tag/*counter_collection.csv 2>/dev/null | head -1)
  echo "== $set"; [ -n "$f" ] && python $R/tools/pmc_summary.py $f 16581375 | grep -i "point_tile" | cut -c1-40,71-160
done
