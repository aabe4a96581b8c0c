#!/bin/bash
# the first VJP stage's tile shape / workgroup size (a build with -DSDFR_VJP_TUNE): duration of vjp_stage_kernel in the C5 loop
for cfg in "2 2 512" "2 2 256" "1 2 256" "1 2 512" "2 4 512" "1 1 256" "4 4 512"; do
  set -- $cfg
  FUSED_SINGLE=5 SDFR_LIB=$PWD/build/variants/libsdfr_vjptune.so SDFR_VJP_TX=$1 SDFR_VJP_TY=$2 SDFR_VJP_THREADS=$3 bash tools/loop_sequence.sh vjptune > /dev/null 2>&1
  echo "TX $1 TY $2 threads $3: $(grep vjp_stage gpurun_out/vjptune/sequence.md) $(tail -1 gpurun_out/vjptune/sequence.md)"
done
