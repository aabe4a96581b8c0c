#!/bin/bash
# pc_variants.sh <variant> ...: the sampler's backward on coherent points (pc_only.py) under several builds
for v in "$@"; do
  lib=$PWD/build/variants/libsdfr_$v.so
  [ "$v" = default ] && lib=$PWD/sdfest_amd/libsdfr_hip.so
  SDFR_LIB=$lib bash tools/trace_cmd.sh pcv_$v tools/microbench/pc_only.py 2>&1 | grep "pc_loss_backward" | sed "s/^/$v /"
done
