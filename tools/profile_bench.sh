#!/bin/bash
# The profiles of a round (through gpurun, from the repo root):  bash tools/profile_bench.sh TAG [WORKLOAD]
#   rocprofv3 kernel trace + stats of the default bench command (BASELINE.json config 3), then the
#   two PMC passes of the same command (separate runs, no trace options combined with --pmc), and
#   the same three passes over the 256^3 smoother measurement (tools/microbench.py: the north-star
#   kernel k_gs_point_tile); summaries under gpurun_out/ -- copy what is to be judged to profiles/.
TAG=${1:-r06}
export PMC_ROUND=$TAG
# options of the 256^3 smoother measurement (default: the coefficient storage the solver uses on such a model)
MB_OPTS=${MB_OPTS:---opt line_compact=1 --opt point_compact=1}
WL=${2:-triaxial256}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
B="python $R/bench.py --workload $WL --no-256 --no-survey --no-cpu-baseline"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o run -- $B > $O/trace.log 2>&1
python $R/tools/rocpd_summary.py $O/trace/run_results.db > $R/gpurun_out/${TAG}_${WL}_kernel_stats.txt
python $R/tools/trace_by_grid.py $O/trace/run_results.db k_line_ > $R/gpurun_out/${TAG}_${WL}_line_launches_by_level.txt
# the bench line of this very run (HIP events under the profiler): its per-launch averages against the table above
grep '^{"metric"' $O/trace.log | tail -1 > $R/gpurun_out/${TAG}_bench_${WL}_profiled_run.json
if [ -n "$ONLY_TRACE" ]; then head -12 $R/gpurun_out/${TAG}_${WL}_kernel_stats.txt; rm -rf $O/trace; exit 0; fi
# (counter collection crashes inside HIP graph replays on this stack: the same kernels, launched eagerly.
#  Even so rocprofv3 segfaults -- once it hung -- in about every second counter pass over this
#  134 000-dispatch run, whatever the library options: bounded time, up to three attempts per pass)
export EMG3D_AMD_GRAPHS=0
BP="$B --steps 3 --warmup 0"
for try in 1 2 3; do
  rm -rf $O/f; timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o run -- $BP > $O/f.log 2>&1 && break
done
for try in 1 2 3; do
  rm -rf $O/w; timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o run -- $BP > $O/w.log 2>&1 && break
done
unset EMG3D_AMD_GRAPHS
F=$(ls $O/f/*counter_collection.csv | head -1); W=$(ls $O/w/*counter_collection.csv | head -1); tail -3 $O/f.log
cd $R && python tools/pmc_traffic.py $WL $F $W > $R/gpurun_out/${TAG}_pmc_traffic_$WL.log 2>&1
python tools/pmc_summary.py $F > $R/gpurun_out/${TAG}_pmc_fetch_$WL.txt
python tools/pmc_summary.py $W > $R/gpurun_out/${TAG}_pmc_write_$WL.txt
# the 256^3 smoothers (point + lines), nu = 2
cd /tmp
M="python $R/tools/microbench.py all --n 256 --slabs \"\""
timeout 600 rocprofv3 --kernel-trace --stats -d $O/mtrace -o run -- python $R/tools/microbench.py all --n 256 --slabs "" --fused-only $MB_OPTS > $O/mtrace.log 2>&1
python $R/tools/rocpd_summary.py $O/mtrace/run_results.db > $R/gpurun_out/${TAG}_smoothers256_kernel_stats.txt
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/mf -o run -- python $R/tools/microbench.py all --n 256 --slabs "" --fused-only $MB_OPTS > $O/mf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/mw -o run -- python $R/tools/microbench.py all --n 256 --slabs "" --fused-only $MB_OPTS > $O/mw.log 2>&1
MF=$(ls $O/mf/*counter_collection.csv | head -1); MW=$(ls $O/mw/*counter_collection.csv | head -1)
cd $R && python tools/pmc_traffic.py smoothers_256 $MF $MW > $R/gpurun_out/${TAG}_pmc_traffic_smoothers256.log 2>&1
python tools/pmc_summary.py $MF 16581375 > $R/gpurun_out/${TAG}_pmc_fetch_smoothers256.txt
python tools/pmc_summary.py $MW 16581375 > $R/gpurun_out/${TAG}_pmc_write_smoothers256.txt
cp $R/profiles/${TAG}_pmc_traffic.json $R/gpurun_out/${TAG}_pmc_traffic.json 2>/dev/null
cat $O/mtrace.log | grep -E "gauss|residual"
rm -rf $O/trace $O/mtrace $O/f $O/w $O/mf $O/mw   # raw traces stay on the box: gpurun_out is capped at 64 MiB
head -30 $R/gpurun_out/${TAG}_${WL}_kernel_stats.txt; tail -5 $R/gpurun_out/${TAG}_pmc_traffic_$WL.log
