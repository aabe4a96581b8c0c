#!/bin/bash
# Run on the GPU box: objects side by side (K = 2 .. 32) for the product build and the variants named on the command
# line (build/variants/libsdfr_<name>.so: SDFR_DIRECT_MIN_TILES / _FEW / SDFR_DIRECT_MIN_LATENTS_FEW builds)
mkdir -p gpurun_out/r05
for v in default "$@"; do
  if [ $v = default ]; then unset SDFR_LIB; else export SDFR_LIB=$PWD/build/variants/libsdfr_$v.so; fi
  echo "== $v"
  python tools/microbench/multi_object_loop.py 2>/dev/null | grep "K=  2\|K=  4\|K=  8\|K= 16\|K= 32" | cut -c1-130
done
