#!/bin/bash
# The profiles of a round (through gpurun, from the repo root):  bash tools/profile_bench.sh TAG
#   kernel trace + stats of the default bench command, then the two PMC passes of the same
#   command (separate runs, no trace options combined with --pmc), summaries under gpurun_out/.
TAG=${1:-v9}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o run -- python $R/bench.py --no-256 --no-survey --no-cpu-baseline > $O.trace.log 2>&1
python $R/tools/rocpd_summary.py $O/trace/run_results.db > $R/gpurun_out/${TAG}_marine128_kernel_stats.txt
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o run -- python $R/bench.py --no-256 --no-survey --no-cpu-baseline > $O.f.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o run -- python $R/bench.py --no-256 --no-survey --no-cpu-baseline > $O.w.log 2>&1
cd $R && python tools/pmc_traffic.py marine128 $O/f/run_counter_collection.csv $O/w/run_counter_collection.csv > $R/gpurun_out/${TAG}_pmc_traffic.log 2>&1
cp $R/profiles/r01_pmc_traffic.json $R/gpurun_out/${TAG}_pmc_traffic.json 2>/dev/null
python tools/pmc_summary.py $O/f/run_counter_collection.csv > $R/gpurun_out/${TAG}_pmc_fetch_marine128.txt
python tools/pmc_summary.py $O/w/run_counter_collection.csv > $R/gpurun_out/${TAG}_pmc_write_marine128.txt
head -12 $R/gpurun_out/${TAG}_marine128_kernel_stats.txt; tail -3 $R/gpurun_out/${TAG}_pmc_traffic.log
