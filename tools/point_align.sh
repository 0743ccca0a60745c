#!/bin/bash
# Does the alignment of the ey / ez rows matter for the tiled point smoother? (through gpurun)
# At nx = 256 the ex rows (256 elements) start on 128-B lines and the ey / ez rows (257 elements = 4112 B) drift
# by 16 B per row; at nx = 255 it is the other way round (ey / ez rows of 256 elements are aligned, ex rows drift);
# nx = 264 / 248: rows of 265 / 249 elements, other drifts. Time per node and HBM bytes per node of k_gs_point_tile.
#   bash tools/point_align.sh      -> gpurun_out/point_align.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/point_align
mkdir -p $O
for shape in 256,256,256 255,256,256 264,256,256 248,256,256 255,255,255; do
  echo "== $shape"
  python $R/tools/microbench.py point --shape $shape --slabs "" 2>&1 | grep -E "tiled|^#"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=$O/${shape//,/x}_$ctr
    rm -rf $d
    timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $d -o run -- python $R/tools/microbench.py point --shape $shape --slabs "" > $d.log 2>&1
    f=$(ls $d/*counter_collection.csv 2>/dev/null | head -1)
    nodes=$(python -c "a=[int(x)-1 for x in '$shape'.split(',')]; print(a[0]*a[1]*a[2])")
    if [ -n "$f" ]; then python $R/tools/pmc_summary.py $f $nodes | grep -i "k_gs_point_tile" | cut -c1-40,71-160; else tail -3 $d.log; fi
    rm -rf $d
  done
done > $R/gpurun_out/point_align.txt 2>&1
cat $R/gpurun_out/point_align.txt
