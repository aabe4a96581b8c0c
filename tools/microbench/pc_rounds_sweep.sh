#!/bin/bash
# Run on the GPU box: the 64-view (and 8-, 1-view) loop's launch sequence, for the build in SDFR_LIB or the product's
mkdir -p gpurun_out/r05
for V in 64 8 1; do
  VIEWS=$V bash tools/loop_sequence.sh r05/loop${V}_groups > gpurun_out/r05/loop${V}_groups.txt 2>&1
  echo "== VIEWS=$V"; grep -h "render_backward_pc_kernel\|launches, kernels" gpurun_out/r05/loop${V}_groups.txt
done
