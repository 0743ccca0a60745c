#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the level-0 line launches at 256^3 and
# 256 x 128 x 128 for one value of a library option (through gpurun):
#   bash tools/pmc_lines256.sh line_stream=1 TAG
OPT=${1:-line_stream=1}
TAG=${2:-x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcl256
for shape in 256,256,256 256,128,128; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=$R/gpurun_out/pmcl256/${TAG}_${shape//,/x}_$ctr
    rm -rf $d
    timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $d -o run -- python $R/tools/microbench.py lines --shape $shape --fused-only --opt $OPT > $d.log 2>&1
    f=$(ls $d/*counter_collection.csv 2>/dev/null | head -1)
    echo "== $OPT $shape $ctr (KiB per launch; FETCH_SIZE x 2 = bytes read)"
    if [ -n "$f" ]; then python $R/tools/pmc_summary.py $f | grep -i "k_line" | cut -c1-48,71-140; else tail -3 $d.log; fi
    rm -rf $d
  done
done
