#!/bin/bash
# Level-0 line passes with several right-hand sides (through gpurun): time per source and HBM traffic per launch
# (FETCH_SIZE / WRITE_SIZE, separate passes) of the batched launches at 256^3, with the batch as a grid dimension
# (line_stream_bmin=0: k_line_colour<BATCH>) and with groups of right-hand sides per factor fetch (k_line_stream).
#   bash tools/batch_lines.sh [shape]      -> gpurun_out/batch_lines/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SHAPE=${1:-256,256,256}
O=$R/gpurun_out/batch_lines
mkdir -p $O
{
echo "## timing (tools/microbench.py lines, per call of 7 launches, per source)"
python $R/tools/microbench.py lines --shape $SHAPE --fused-only --batch 1
python $R/tools/microbench.py lines --shape $SHAPE --fused-only --batch 1 --opt line_stream=3
for B in 2 4; do
  python $R/tools/microbench.py lines --shape $SHAPE --fused-only --batch $B --opt line_stream_bmin=0
  python $R/tools/microbench.py lines --shape $SHAPE --fused-only --batch $B
done
} > $O/timing_${SHAPE//,/x}.txt 2>&1
cat $O/timing_${SHAPE//,/x}.txt | grep -v amdgpu.ids
if [ -n "$TIMING_ONLY" ]; then exit 0; fi
for cfg in "1 line_stream=1" "1 line_stream=3" "2 line_stream_bmin=0" "2 line_stream_bmin=64" "4 line_stream_bmin=0" "4 line_stream_bmin=64"; do
  set -- $cfg
  B=$1; OPT=$2
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=$O/pmc_B${B}_${OPT//=/}_$ctr
    rm -rf $d
    timeout 400 rocprofv3 --pmc $ctr --output-format csv -d $d -o run -- python $R/tools/microbench.py lines --shape $SHAPE --fused-only --batch $B --opt $OPT > $d.log 2>&1
    f=$(ls $d/*counter_collection.csv 2>/dev/null | head -1)
    echo "== batch $B $OPT $SHAPE $ctr (KiB per launch; FETCH_SIZE x 2 = bytes read)"
    if [ -n "$f" ]; then python $R/tools/pmc_summary.py $f | grep -i "k_line" | cut -c1-64,71-140; else tail -3 $d.log; fi
    rm -rf $d
  done
done > $O/pmc_${SHAPE//,/x}.txt 2>&1
