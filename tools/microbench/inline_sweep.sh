#!/bin/bash
# Run on the GPU box: the 8-object loop and the 8-view loop of one object for builds with other SDFR_INLINE_MAX_VIEWS
mkdir -p gpurun_out/r05
for v in default "$@"; do
  if [ $v = default ]; then unset SDFR_LIB; else export SDFR_LIB=$PWD/build/variants/libsdfr_$v.so; fi
  echo "== $v"
  python tools/microbench/multi_object_loop.py 2>/dev/null | grep "K=  4\|K=  8" | cut -c1-130
  python tools/microbench/loop_defer.py 2>/dev/null | grep "views   4\|views   8\|views  16"
done
