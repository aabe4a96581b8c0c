#!/bin/bash
# Run on the GPU box: the plain and the loss-fused step's kernels (tools/microbench/step_l1_kernels.py under the kernel
# trace) for the product build and the variants named on the command line (build/variants/libsdfr_<name>.so)
mkdir -p gpurun_out/r05
export SDFR_DEFER=0
for v in default "$@"; do
  if [ $v = default ]; then unset SDFR_LIB; else export SDFR_LIB=$PWD/build/variants/libsdfr_$v.so; fi
  bash tools/trace_full.sh r05/l1_$v tools/microbench/step_l1_kernels.py both 2>&1 | grep "render_\|loss_reduce\|us per step" | grep -v "calls     [12] " | cut -c1-200
done
