#!/bin/bash
# loop_variants.sh <variant> ...: C5 ms per iteration (tools/bench_loop.py, three runs each) under each build variant
for v in "$@"; do
  lib=$PWD/build/variants/libsdfr_$v.so
  [ "$v" = default ] && lib=$PWD/sdfest_amd/libsdfr_hip.so
  for r in 1 2 3; do
    echo -n "$v: "; SDFR_LIB=$lib python tools/bench_loop.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_iteration_graph'], d['graph_final_position_error_mm'])"
  done
done
