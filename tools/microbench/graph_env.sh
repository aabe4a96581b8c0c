#!/bin/bash
# graph_env.sh: tools/microbench/graph_env.py under each of the runtime's graph switches, one process per setting
out=${1:-gpurun_out/graph_env.txt}
mkdir -p "$(dirname "$out")"
: > "$out"
run() { echo "== $*" | tee -a "$out"; env "$@" python tools/microbench/graph_env.py 2>/dev/null | tail -1 | tee -a "$out"; }
run X=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=64
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run HIP_FORCE_DEV_KERNARG=1
run GPU_MAX_HW_QUEUES=1
run X=0
