# A/B of HIP runtime knobs around graph replay and dispatch on the cycle time of the default bench workload (through
# gpurun: bash tools/env_ab.sh; round 3: profiles/r03_runtime_knobs.txt -- none of them matters, nor do the graphs at 256^3)
B="python bench.py --no-256 --no-survey --no-cpu-baseline --no-ttt --steps 20 --warmup 3"
run() { echo -n "$1 : "; env $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2))"; }
run X=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=0
run DEBUG_HIP_GRAPH_BATCH_SIZE=16
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run ROC_AQL_QUEUE_SIZE=65536
run DEBUG_CLR_MAX_BATCH_SIZE=4096
run EMG3D_AMD_GRAPHS=0
run X=2
