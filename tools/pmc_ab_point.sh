cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r2d
for pf in 0 3; do
for set in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/r2d/pmc_${pf}_$set -o run -- python $R/tools/microbench.py point --n 256 --slabs "" --opt point_prefetch=$pf > $R/gpurun_out/r2d/pmc_${pf}_$set.log 2>&1
  f=$(ls $R/gpurun_out/r2d/pmc_${pf}_$set/*counter_collection.csv 2>/dev/null | head -1)
  echo "== pf=$pf $set"; [ -n "$f" ] && python $R/tools/pmc_summary.py $f 16581375 | grep -i "point_tile" | cut -c1-40,71-160
done; done
