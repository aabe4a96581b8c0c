#!/bin/bash
# Run on the GPU box: objects side by side (K = 4 .. 32) for the product build and the SDFR_DIRECT_MIN_TILES variants
mkdir -p gpurun_out/r05
for v in default dmt240 dmt128; do
  if [ $v = default ]; then unset SDFR_LIB; else export SDFR_LIB=$PWD/build/variants/libsdfr_$v.so; fi
  echo "== $v"
  python tools/microbench/multi_object_loop.py 2>/dev/null | grep "K=  4\|K=  8\|K= 16\|K= 32" | cut -c1-130
done
