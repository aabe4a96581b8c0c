#!/bin/bash
# bt_variants.sh <tag> <variant> ...: the transposed-resize launches of a batched decoder VJP under each build variant
tag=$1; shift
for v in "$@"; do
  lib=$PWD/build/variants/libsdfr_$v.so
  [ "$v" = default ] && lib=$PWD/sdfest_amd/libsdfr_hip.so
  SDFR_LIB=$lib bash tools/trace_cmd.sh ${tag}_$v tools/profile_decoder_vjp.py > /dev/null 2>&1
  python tools/microbench/bt_calls.py gpurun_out/${tag}_$v
done
